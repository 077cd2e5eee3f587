// map_update.h -- log-odds map update (updateByScan) and probability-texel maintenance
// as gfx950 HIP kernels.
//
// Reference: OccGridMapBase::updateByScan + updateLineBresenhami / bresenham2D /
// bresenhamCellFree / bresenhamCellOcc, HSL/map/OccGridMapBase.h:121-260, cell rules
// GridMapLogOdds.h:135-156.  Net effect per scan and per cell:
//   * a cell that is the END cell of any non-skipped beam gets the occupied update,
//     at most once; otherwise a cell crossed by any beam gets the free update, once;
//   * if (in beam order) a free touch came BEFORE the first occupied touch, the
//     reference first applies and then reverts the free update, so the float result
//     is ((l + f) - f) [+ o] instead of l [+ o].  That rounding artefact is reproduced.
//
// Parallel formulation (bit-exact with the sequential reference):
//   pass 1 "mark" (per beam), two launches: (a) every beam atomicMax'es
//       key = (scan serial << 20) | (0xFFFFF - beam index) into the occ-key plane at its end cell,
//       which leaves the FIRST beam (lowest index) ending there; (b) one wavefront per beam, lane k
//       owns Bresenham steps k, k+64, ... -- the cell of step i has a closed form (minor steps =
//       floor((e0 + i*db)/da)), so no lane walks the line sequentially -- and tags every crossed
//       cell in the free-key plane: a plain store of the serial where no beam ends, atomicMax of
//       the key (lowest crossing beam index, needed for the revert artefact) where one does.
//   pass 2 "apply" (per cell, DENSE over the bounding box of the scan):  every cell whose keys
//       carry the current serial applies the reference's rule -- occupied if any beam ends there
//       (after undoing the free update when a lower-indexed beam crossed it first), else free --
//       to the log-odds plane, writes the reference's updateIndex stamp and the new probability.
//       Rows are contiguous, so all loads and stores are coalesced; no float atomics, no races.
//   pass 3 "texels" (per cell, dense over the box grown by one):  rebuilds the float4 texels
//       {P(x,y),P(x+1,y),P(x,y+1),P(x+1,y+1)} the matcher samples.
// Keys of earlier scans are always smaller than the current ones, so the key planes never need
// clearing (only when the 12-bit serial wraps, every 4095 updates).
//
// Traffic: mark = one 4-byte load (+ rarely an atomic) per visited cell; apply = 12 B
// read + 12 B written per touched cell of the box; texels = 16 B written per cell of the box.
// HBM/L2-bound integer/byte work, no MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include "gn_match.h"

namespace hsm {

// read-write view of one level for the update path
struct LevelRW {
  float* logodds;         // LogOddsCell::logOddsVal plane
  int* update_index;      // LogOddsCell::updateIndex plane
  float* prob;            // p = e^l / (e^l + 1)
  float4* quad;           // {P(x,y), P(x+1,y), P(x,y+1), P(x+1,y+1)}
  unsigned int* key_free; // first free-touching beam of the current scan
  unsigned int* key_occ;  // first end-cell beam of the current scan
  unsigned int* occ_bits; // 1 bit per cell: "some beam of the current scan ends here" (set in pass 1a, cleared in pass 2)
  unsigned char* free_bytes; // dense scans: 1 byte per cell "some beam of the current scan crosses this cell", in tiles of 16 x 8
                             // cells (index = mark_index: a tile is one 128-byte line); set by update_mark_free_dense_kernel,
                             // cleared by the dense apply pass
  int sx, sy;
  int tiles_x, quad_texels;  // tiled texel plane geometry (gn_match.h quad_index)
  int kf_tiles_x;            // free-key tiles per row = key_free_tiles_x(sx): ceil(sx / 64) * 8   (key_free_index)
};

// The free-key plane is stored in 8x4-cell tiles (= one 128-byte line).  The line walk of update_mark_free_kernel
// writes one 4-byte tag per visited cell: row major, a y-major beam touches a new cache line every step and an
// x-major one every 32 steps; tiled, both touch a new line every 4..8 steps, and the lanes of a wave (64
// consecutive steps of one beam) share lines either way.  HSM_KEYFREE_TILE=0: row major.
#ifndef HSM_KEYFREE_TILE
#define HSM_KEYFREE_TILE 1
#endif
// tiles per tile row, padded to whole 64-cell BLOCKS (8 tiles): the dense apply pass owns the marks of a 64 x 4-cell block as
// 256 CONTIGUOUS bytes, so the last block of a row must not run into the next tile row -- with the padding the dense form
// works for every map width (round 4; until then rows had to be a multiple of 64 cells)
__host__ __device__ __forceinline__ int key_free_tiles_x(int sx) { return ((sx + 63) / 64) * 8; }
__host__ __device__ __forceinline__ size_t key_free_cells(int sx, int sy) {
#if HSM_KEYFREE_TILE
  return (size_t)key_free_tiles_x(sx) * (size_t)((sy + 3) / 4) * 32u;
#else
  return (size_t)sx * sy;
#endif
}
__device__ __forceinline__ unsigned int key_free_index(const LevelRW& L, unsigned int x, unsigned int y) {
#if HSM_KEYFREE_TILE
  return ((((y >> 2) * (unsigned int)L.kf_tiles_x) + (x >> 3)) << 5) | ((y & 3u) << 3) | (x & 7u);
#else
  return y * (unsigned int)L.sx + x;
#endif
}

// The mark BYTES of the dense form (free_bytes): tiles of 16 x 8 cells = one 128-byte line each (HSM_MARK_TILE16=1, the default
// since the end of round 4).  Until then they shared the free-key plane's 8 x 4-cell tiling (HSM_MARK_TILE16=0), in which a
// 128-byte line of BYTES is four tiles side by side = 32 x 4 cells: the 64 steps of a line-walk iteration cross about
// (dx / 32 + dy / 4 + 1) lines -- ~14.5 averaged over the beam directions of a 360-degree scan -- against (dx / 16 + dy / 8 + 1)
// ~ 10 with the squarer tile.  The line walk is bound by exactly these scattered byte accesses (profiles/r04/README.md 5):
// 65.2 -> 57.6 us on configs[4]; the apply pass, which now owns 32 x 8-cell blocks (two tiles = 256 contiguous mark bytes, its
// plane accesses two 128-byte row segments per wavefront): 68.4 -> 65.1 us; update 0.144 -> 0.1355 ms, maps bit-identical
// (profiles/r04/README.md 20).
#ifndef HSM_MARK_TILE16
#define HSM_MARK_TILE16 1
#endif
__host__ __device__ __forceinline__ int mark_tiles_x(int sx) { return ((sx + 31) / 32) * 2; }  // 16-cell tiles per row, whole 32-cell blocks
__host__ __device__ __forceinline__ size_t mark_bytes(int sx, int sy) {
#if HSM_MARK_TILE16
  return (size_t)mark_tiles_x(sx) * (size_t)((sy + 7) / 8) * 128u;
#else
  return key_free_cells(sx, sy);
#endif
}
// One byte per 16 x 8 mark TILE behind the mark bytes: "a beam of the current scan ends in this tile" (set by the end-cell pass,
// cleared by the apply pass; HSM_MARK_TILE_END=1, the default since the end of round 4).  The line walk had to read every mark
// byte before storing to it -- to learn whether a beam ends in the cell (then the keyed atomicMax decides the revert artefact),
// and to skip marks already set; that load, ~10 lines of a 67 MB plane per iteration, was a third of the walk.  Now it reads the
// TILE's byte -- a 128 x smaller, cache-resident map, 1-4 lines per iteration -- and only in the ~1/6 of the tiles where it is set
// the cell's own byte; everywhere else it stores its mark unread (an already set mark is stored again: same value).
// configs[4]: line walk 57.5 -> 50.0 us, update 0.135 -> 0.127 ms (profiles/r04/README.md 21).
#ifndef HSM_MARK_TILE_END
#define HSM_MARK_TILE_END 1
#endif
#if HSM_MARK_TILE_END && !HSM_MARK_TILE16
#error "the tile flags index the 16 x 8 mark tiles"
#endif
__host__ __device__ __forceinline__ size_t mark_tile_end_offset(int sx, int sy) { return mark_bytes(sx, sy) + 256; }
__host__ __device__ __forceinline__ size_t mark_plane_bytes(int sx, int sy) {
  return mark_bytes(sx, sy) + 256 + (HSM_MARK_TILE_END ? ((mark_bytes(sx, sy) / 128 + 3) & ~(size_t)3) + 256 : 0);
}
__device__ __forceinline__ unsigned int mark_index(const LevelRW& L, unsigned int x, unsigned int y) {
#if HSM_MARK_TILE16
  return ((((y >> 3) * (unsigned int)mark_tiles_x(L.sx)) + (x >> 4)) << 7) | ((y & 7u) << 4) | (x & 15u);
#else
  return key_free_index(L, x, y);
#endif
}

// Key = (generation of the scan << kBeamBits) | (kBeamMask - beam index): atomicMax keeps the newest scan and, within
// it, the LOWEST beam index.  20 bits of beam index (scans of up to 1 048 575 beams; the reference has no limit, and
// neither has any sensor), 12 bits of generation: the key planes are cleared once every 4095 updates of a level.
constexpr unsigned int kBeamBits = 20;
constexpr unsigned int kBeamMask = (1u << kBeamBits) - 1u;
constexpr unsigned int kSerialMax = (1u << (32 - kBeamBits)) - 1u;

// What the dense line walk needs to know about ONE beam on ONE level -- the result of beam_line() and of the two divisions
// of its per-64-steps increment.  Round 3's walk derived all of this per WAVEFRONT (one beam each, 64 lanes computing the same
// numbers: ~300 of the kernel's VALU instructions per beam, twice -- the beam and its predecessor --, half of its 30 M);
// since round 4 the end-cell pass, which evaluates beam_line() per LANE anyway, stores the record and the walk reads two of
// them through the scalar cache (wave-uniform address: s_load_dwordx4 into SGPRs).
struct BeamRec {
  unsigned int abs_da;  // major-axis steps = free cells of the line; 0 = the beam is skipped (off the map, begin == end, NaN)
  unsigned int abs_db;
  unsigned int q64;     // (64 * abs_db) / abs_da: minor steps per 64 major steps ...
  unsigned int r64_oct; // ... and the remainder, << 3 | octant: bit 0 x is the major axis, bit 1 major step > 0, bit 2 minor step > 0
};

struct UpdateParams {
  LevelRW lv;
  Affine2 pose;           // Translation(mapPose.xy) * Rotation(mapPose.theta), host sinf/cosf
  const float2* pts;      // level-0 endpoints (robot frame)
  int n;
  float pt_scale;         // 2^-level (exact), DataPointContainer.h:46-58
  int bx, by;             // scanBeginMapi (OccGridMapBase.h:137)
  unsigned int serial;    // 1..kSerialMax
  float log_odds_free, log_odds_occ;
  int mark_free, mark_occ;  // currMarkFreeIndex / currMarkOccIndex (OccGridMapBase.h:123-124)
  int x0, y0, x1, y1;       // inclusive cell bounding box of everything this scan can touch
  struct BeamRec* recs;     // dense scans: one record per beam of this level (update_mark_occ_dense_kernel -> the line walk)
};

// GridMapLogOddsFunctions::getGridProbability (GridMapLogOdds.h:163-166): exp(float) is glibc's expf
// there; libm_exact.h reproduces it bit for bit
__device__ __forceinline__ float grid_probability(float log_odds) {
  const float odds = libm::expf_glibc(log_odds);
  return odds / (odds + 1.0f);
}

// All levels of one updateByScan in ONE launch per pass: blockIdx.y selects the level (the levels are
// independent maps, so they run concurrently and the small coarse levels hide behind level 0).
struct UpdateBatch {
  UpdateParams lv[kMaxLevels];
  int nlev;
};

struct BeamLine {
  bool valid;
  int x1, y1;
  unsigned int abs_da, abs_db;
  int offset_a, offset_b;
  unsigned int e0;
  unsigned int start;
  bool x_major;  // the major (per-step) axis is x
};

// geometry of beam i exactly as updateByScan / updateLineBresenhami derive it
__device__ __forceinline__ BeamLine beam_line(const UpdateParams& P, int i) {
  BeamLine b;
  const float2 p = P.pts[i];
  float ex, ey;
  affine_apply(P.pose, p.x * P.pt_scale, p.y * P.pt_scale, ex, ey);  // OccGridMapBase.h:148
  ex += 0.5f;                                                         // :151
  ey += 0.5f;
  b.x1 = (int)ex;  // cast<int>() truncation, :154
  b.y1 = (int)ey;
  const int x0 = P.bx, y0 = P.by;
  b.valid = !(x0 == b.x1 && y0 == b.y1);  // :158
  // A NaN endpoint: x86's cvttss2si makes it INT_MIN, which fails the bounds test below and drops the beam; this
  // device's conversion makes it 0, a VALID cell.  The same finite-range test the host's bounding box uses
  // (update_level) keeps both consistent with the reference; every finite coordinate it rejects fails :176-188 anyway.
  if (!(ex > -2.0f && ex < (float)P.lv.sx + 2.0f && ey > -2.0f && ey < (float)P.lv.sy + 2.0f)) b.valid = false;
  // both endpoints inside the map, :176-188
  if ((x0 < 0) || (x0 >= P.lv.sx) || (y0 < 0) || (y0 >= P.lv.sy)) b.valid = false;
  if ((b.x1 < 0) || (b.x1 >= P.lv.sx) || (b.y1 < 0) || (b.y1 >= P.lv.sy)) b.valid = false;
  const int dx = b.x1 - x0;
  const int dy = b.y1 - y0;
  const unsigned int abs_dx = (unsigned int)(dx < 0 ? -dx : dx);
  const unsigned int abs_dy = (unsigned int)(dy < 0 ? -dy : dy);
  const int offset_dx = dx > 0 ? 1 : -1;                 // util::sign, sign(0) = -1
  const int offset_dy = (dy > 0 ? 1 : -1) * P.lv.sx;
  b.start = (unsigned int)(y0 * P.lv.sx + x0);
  b.x_major = abs_dx >= abs_dy;
  if (abs_dx >= abs_dy) {  // :200-207
    b.abs_da = abs_dx;
    b.abs_db = abs_dy;
    b.offset_a = offset_dx;
    b.offset_b = offset_dy;
  } else {
    b.abs_da = abs_dy;
    b.abs_db = abs_dx;
    b.offset_a = offset_dy;
    b.offset_b = offset_dx;
  }
  b.e0 = b.abs_da / 2;
  return b;
}

// cell visited at Bresenham step i (0 = start cell), closed form of bresenham2D (:243-260):
// after i major steps the error accumulator has crossed abs_da floor((e0 + i*db)/da) times.
__device__ __forceinline__ unsigned int line_cell(const BeamLine& b, unsigned int i) {
  const unsigned int minor = (b.e0 + i * b.abs_db) / b.abs_da;
  return b.start + (unsigned int)((int)i * b.offset_a) + (unsigned int)((int)minor * b.offset_b);
}

// pass 1a: end cells.  One thread per beam: atomicMax leaves the FIRST beam that ends in a cell.
// Neighbouring beams end in the same cell (near walls) or in the same 32-cell bitmap word (walls along x), and
// same-address atomics serialise in L2, so each wavefront first combines what it can: of a run of adjacent lanes
// with the same end cell only the first (lowest beam index = largest key, exactly what atomicMax would keep)
// issues the atomicMax, and the bits of a run of adjacent lanes with the same bitmap word are OR-ed by a
// segmented scan so that only the first lane of the run issues the atomicOr.
__device__ __forceinline__ void mark_occ_block(const UpdateParams& P, unsigned int block) {
  const int beam = block * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (((int)(block * blockDim.x) + (int)(threadIdx.x & ~63u)) >= P.n) return;  // whole wave beyond the scan
  bool valid = beam < P.n;
  unsigned int c = 0xffffffffu;
  if (valid) {
    const BeamLine b = beam_line(P, beam);
    valid = b.valid;
    if (valid) c = (unsigned int)(b.y1 * P.lv.sx + b.x1);
  }
  const unsigned int c_prev = (unsigned int)__shfl_up((int)c, 1);
  const bool first_of_cell = valid && (lane == 0 || c_prev != c);
  if (first_of_cell) atomicMax(&P.lv.key_occ[c], (P.serial << kBeamBits) | (kBeamMask - (unsigned int)beam));
  // bitmap: runs of adjacent valid lanes with the same word
  const unsigned int w = valid ? (c >> 5) : (0xfffffff0u - (unsigned int)lane);  // invalid lanes never join a run
  const unsigned int w_prev = (unsigned int)__shfl_up((int)w, 1);
  const bool head = lane == 0 || w_prev != w;
  const unsigned long long heads = __ballot(head);
  unsigned int m = valid ? (1u << (c & 31u)) : 0u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned int up = (unsigned int)__shfl_down((int)m, d);
    // lanes lane+1 .. lane+d belong to this lane's run iff none of them starts a new one
    const bool same = (lane + d < 64) && (((heads >> (lane + 1)) & ((1ull << d) - 1ull)) == 0ull);
    if (same) m |= up;
  }
  if (head && valid) atomicOr(&P.lv.occ_bits[w], m);
}

#if defined(HSM_EXPERIMENTS)  // rounds 1-2: dense scans on the keyed planes in two launches (superseded by the byte-map form)
__global__ void __launch_bounds__(256) update_mark_occ_kernel(const UpdateBatch B) {
  mark_occ_block(B.lv[blockIdx.y], blockIdx.x);
}
#endif

// dense scans: "a beam ends here" is bit 1 of the cell's mark byte instead of a bit of the row-major end-cell bitmap
constexpr unsigned char kMarkCrossed = 1, kMarkEnd = 2;

// pass 1a of a dense scan: the occ key as above; the end-cell flag goes into the cell's MARK BYTE (value 2)
// -- the byte the line walk of pass 1b reads and writes anyway, in the same tiled plane -- and the bitmap is not touched.
// All writers of a byte store the same value; the launch boundary orders them before pass 1b.
__global__ void __launch_bounds__(256) update_mark_occ_dense_kernel(const UpdateBatch B) {
  const UpdateParams& P = B.lv[blockIdx.y];
  const int beam = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  if (((int)(blockIdx.x * blockDim.x) + (int)(threadIdx.x & ~63u)) >= P.n) return;  // whole wave beyond the scan
  bool valid = beam < P.n;
  unsigned int c = 0xffffffffu, kc = 0u;
  if (valid) {
    const BeamLine b = beam_line(P, beam);
    valid = b.valid;
    BeamRec rec = {0u, 0u, 0u, 0u};
    if (valid) {
      c = (unsigned int)(b.y1 * P.lv.sx + b.x1);
      kc = mark_index(P.lv, (unsigned int)b.x1, (unsigned int)b.y1);
      // the walk's per-64-steps increment of (e0 + i * db) / da: quotient and remainder (abs_da >= 1 for a valid beam)
      const unsigned int inc = 64u * b.abs_db;
      const unsigned int q64 = inc / b.abs_da;
      const bool major_pos = b.offset_a > 0, minor_pos = b.offset_b > 0;
      rec.abs_da = b.abs_da;
      rec.abs_db = b.abs_db;
      rec.q64 = q64;
      rec.r64_oct = ((inc - q64 * b.abs_da) << 3) | (b.x_major ? 1u : 0u) | (major_pos ? 2u : 0u) | (minor_pos ? 4u : 0u);
    }
    reinterpret_cast<uint4*>(P.recs)[beam] = make_uint4(rec.abs_da, rec.abs_db, rec.q64, rec.r64_oct);
  }
  const unsigned int c_prev = (unsigned int)__shfl_up((int)c, 1);
  if (valid && (lane == 0 || c_prev != c)) {  // of a run of adjacent lanes with the same end cell the first = lowest beam index
    atomicMax(&P.lv.key_occ[c], (P.serial << kBeamBits) | (kBeamMask - (unsigned int)beam));
    P.lv.free_bytes[kc] = kMarkEnd;
#if HSM_MARK_TILE_END
    P.lv.free_bytes[mark_tile_end_offset(P.lv.sx, P.lv.sy) + (kc >> 7)] = 1;
#endif
  }
}

// pass 1b: line cells (after 1a has completed).  WHICH beam crossed a cell first only matters
// where some beam also ENDS (the free-then-occupied revert, OccGridMapBase.h:231-233); everywhere
// else "some beam of this scan crossed it" is all the apply pass needs.  So a cell that is nobody's
// end cell gets a plain store of the bare serial tag (all writers store the same word: a benign
// race, and stores do not serialise in L2 the way same-address atomics do next to the sensor),
// and only end cells -- a few thousand per scan -- take the atomicMax that keeps the lowest beam
// index.  A cell is classified by the end-cell bitmap, which pass 1a finalised, so the two kinds of
// access never mix on one word.
//
// KEYED = true (scans below 4096 beams, where the whole update is launch-latency bound): every crossed cell takes the
// atomicMax of the full key, end cell or not, so the pass does not read the end-cell bitmap and no longer depends on
// pass 1a -- both run in ONE launch (update_mark_kernel) and the update is one dependent launch shorter.  The apply
// pass reads the same information either way (the serial tag; the beam index only where a beam ends).
template <bool KEYED>
__device__ __forceinline__ void mark_free_block(const UpdateParams& P, unsigned int block) {
  const int lane = threadIdx.x & 63;
  const int beam = block * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (beam >= P.n) return;
  const BeamLine b = beam_line(P, beam);
  if (!b.valid) return;
  const unsigned int tag = P.serial << kBeamBits;
  const unsigned int key = tag | (kBeamMask - (unsigned int)beam);
  if ((unsigned int)lane >= b.abs_da) return;
  // Lane k visits steps k, k+64, ...: instead of one integer division per step (line_cell), carry the
  // quotient/remainder of (e0 + i*db) / da forward by the per-64-step increment -- two divisions per lane.
  const unsigned int num0 = b.e0 + (unsigned int)lane * b.abs_db;
  unsigned int q = num0 / b.abs_da, r = num0 - q * b.abs_da;
  const unsigned int inc = 64u * b.abs_db;
  const unsigned int q64 = inc / b.abs_da, r64 = inc - q64 * b.abs_da;
  const unsigned int step_a = (unsigned int)(64 * b.offset_a);
  unsigned int base = b.start + (unsigned int)(lane * b.offset_a);
  const bool x_major = b.x_major;
  // Duplicate suppression.  Neighbouring beams of a dense scan run through the SAME cells for their first
  // hundreds of steps (all lines start in the same cell and separate by less than a cell until 1/dtheta
  // cells out).  If beam-1 is valid, lies in the same octant and visits the same cell at step i, it writes
  // the same tag there -- or, on an end cell, a LARGER key (lower beam index) -- so this beam's access is
  // redundant and is skipped; by induction the lowest-indexed beam of each run does the write.
  const BeamLine pb = beam > 0 ? beam_line(P, beam - 1) : b;
  const bool dedup = beam > 0 && pb.valid && pb.offset_a == b.offset_a && pb.offset_b == b.offset_b;
  const unsigned int pnum0 = pb.e0 + (unsigned int)lane * pb.abs_db;
  unsigned int pq = dedup ? pnum0 / pb.abs_da : 0u, pr = dedup ? pnum0 - pq * pb.abs_da : 0u;
  const unsigned int pinc = 64u * pb.abs_db;
  const unsigned int pq64 = dedup ? pinc / pb.abs_da : 0u, pr64 = dedup ? pinc - pq64 * pb.abs_da : 0u;
  for (unsigned int i = lane; i < b.abs_da; i += 64) {  // abs_da free cells: steps 0 .. abs_da-1
    if (!(dedup && i < pb.abs_da && pq == q)) {
      const unsigned int c = base + (unsigned int)((int)q * b.offset_b);  // == line_cell(b, i)
      // the same cell as (x, y): i steps along the major axis, q along the minor one
      const int sa = b.offset_a > 0 ? (int)i : -(int)i, sb = b.offset_b > 0 ? (int)q : -(int)q;
      const unsigned int cx = (unsigned int)(P.bx + (x_major ? sa : sb)), cy = (unsigned int)(P.by + (x_major ? sb : sa));
      const unsigned int kc = key_free_index(P.lv, cx, cy);
      if (KEYED || ((P.lv.occ_bits[c >> 5] >> (c & 31u)) & 1u)) {  // (the bitmap is 32x denser than the key plane: stays in L2)
        atomicMax(&P.lv.key_free[kc], key);
      } else {
        P.lv.key_free[kc] = tag;
      }
    }
    base += step_a;
    q += q64;
    r += r64;
    if (r >= b.abs_da) {
      r -= b.abs_da;
      ++q;
    }
    pq += pq64;
    pr += pr64;
    if (dedup && pr >= pb.abs_da) {
      pr -= pb.abs_da;
      ++pq;
    }
  }
}

#if defined(HSM_EXPERIMENTS)
__global__ void __launch_bounds__(256) update_mark_free_kernel(const UpdateBatch B) {
  mark_free_block<false>(B.lv[blockIdx.y], blockIdx.x);
}
#endif

// passes 1a + 1b of a SMALL scan in one launch: the first occ_blocks workgroups of a row mark the end cells, the rest
// walk the lines with keyed atomics (no dependency between the two, see mark_free_block)
__global__ void __launch_bounds__(256) update_mark_kernel(const UpdateBatch B, unsigned int occ_blocks) {
  const UpdateParams& P = B.lv[blockIdx.y];
  if (blockIdx.x < occ_blocks)
    mark_occ_block(P, blockIdx.x);
  else
    mark_free_block<true>(P, blockIdx.x - occ_blocks);
}

// dense over the box [x0..x1] x [y0..y1]: bresenhamCellFree / bresenhamCellOcc (OccGridMapBase.h:216-241)
//
// "A beam of THIS scan ends here" is the cell's bit in the end-cell bitmap (set by update_mark_occ_kernel,
// cleared here), so the occ-key plane is only read where the bit is set -- end cells are walls, and most of that
// plane's cache lines are never touched by this pass.  The bitmap word is cleared by the lane of its first cell;
// that is only safe when every reader of a word sits in the SAME wavefront (reads program-ordered before the
// store), which holds when rows are a multiple of 64 cells: the pass then runs over the box widened to 64-cell
// column boundaries (the extra cells carry no key of this scan).  Other map widths keep the plain form.
// SCATTER_TEXELS (quad layout): the cell's new probability goes straight into the four texels it is a
// corner of -- component 0 of texel (x,y), 1 of (x-1,y), 2 of (x,y-1), 3 of (x-1,y-1), with update_texels_kernel's
// edge replication at the last column / row -- so the texel pass and its launch disappear.  Every texel component
// has exactly one writer (the thread of its cell), untouched components keep their value: same bits as the rebuild.
template <bool SCATTER_TEXELS>
__global__ void __launch_bounds__(256) update_apply_kernel(const UpdateBatch B) {
  const UpdateParams& P = B.lv[blockIdx.y];
  if (P.x1 < P.x0) return;  // this level has nothing to apply
  const bool aligned = (P.lv.sx & 63) == 0;
  const int bx0 = aligned ? (P.x0 & ~63) : P.x0;
  const int w = aligned ? ((P.x1 | 63) - bx0 + 1) : (P.x1 - P.x0 + 1), h = P.y1 - P.y0 + 1;
  const size_t n = (size_t)w * h;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int x = bx0 + (int)(t % (size_t)w), y = P.y0 + (int)(t / (size_t)w);
    const size_t c = (size_t)y * P.lv.sx + x;
    const unsigned int kf = P.lv.key_free[key_free_index(P.lv, (unsigned int)x, (unsigned int)y)];
    unsigned int ko;
    bool occ;
    if (aligned) {
      const unsigned int word = P.lv.occ_bits[c >> 5];
      occ = (word >> (c & 31u)) & 1u;
      ko = occ ? P.lv.key_occ[c] : 0u;
      occ = occ && (ko >> kBeamBits) == P.serial;  // (a bit without this scan's key cannot occur; cheap to insist)
      if (word != 0u && (c & 31u) == 0) P.lv.occ_bits[c >> 5] = 0u;
    } else {
      // every set bit lies inside the box, so zeroing each word that overlaps it is exact
      if ((c & 31u) == 0 || x == P.x0) P.lv.occ_bits[c >> 5] = 0u;
      ko = P.lv.key_occ[c];
      occ = (ko >> kBeamBits) == P.serial;
    }
    const bool fre = (kf >> kBeamBits) == P.serial;
    if (!fre && !occ) continue;
    float l = P.lv.logodds[c];
    int stamp;
    if (occ) {
      // free-touched by an earlier beam of this scan: applied, then reverted (:231-233)
      if (fre && (kBeamMask - (kf & kBeamMask)) < (kBeamMask - (ko & kBeamMask))) {
        l += P.log_odds_free;
        l -= P.log_odds_free;
      }
      if (l < 50.0f) l += P.log_odds_occ;  // updateSetOccupied
      stamp = P.mark_occ;
    } else {
      l += P.log_odds_free;                // updateSetFree
      stamp = P.mark_free;
    }
    P.lv.logodds[c] = l;
    P.lv.update_index[c] = stamp;
    const float p = grid_probability(l);
    P.lv.prob[c] = p;
    if (SCATTER_TEXELS) {
      float* q = reinterpret_cast<float*>(P.lv.quad);
      const int sx = P.lv.sx, sy = P.lv.sy;
      const bool lastx = x == sx - 1, lasty = y == sy - 1;
      auto put = [&](int tx, int ty, int comp) { q[4 * (size_t)quad_index(tx, ty, P.lv.tiles_x, sx) + comp] = p; };
      put(x, y, 0);
      if (x > 0) put(x - 1, y, 1);
      if (lastx) put(x, y, 1);
      if (y > 0) put(x, y - 1, 2);
      if (lasty) put(x, y, 2);
      if (x > 0 && y > 0) put(x - 1, y - 1, 3);
      if (lastx && y > 0) put(x, y - 1, 3);
      if (lasty && x > 0) put(x - 1, y, 3);
      if (lastx && lasty) put(x, y, 3);
    }
  }
}


// ---- dense scans (>= HSM_MERGED_MARK_MAX beams): one BYTE per crossed cell instead of a key -------------------------------
// Measured on the 16 k-beam scans of configs[4] (profiles/r03/README.md): the keyed form moves 4.7x the algorithmic bytes
// -- the line walk writes a 4-byte tag per crossed cell and the dense apply pass reads a 4-byte key for EVERY cell of the
// bounding box (2x the touched cells).  WHICH beam crossed a cell first only matters where a beam also ends (see
// mark_free_block); everywhere else "crossed by this scan" is all there is to say.
//   update_mark_occ_dense_kernel: the occ key of an end cell, and the value 2 in its mark byte ("a beam ends here").
//   update_mark_free_dense_kernel: mark_free_block's walk (one wavefront per beam, lane k owns steps k, k+64, ...; duplicate
//     suppression against the previous beam) on ONE byte per crossed cell -- a quarter of the bytes, and consecutive lanes
//     touch consecutive bytes of a tile row.  A step loads the byte: flagged as an end cell, it takes the keyed atomicMax as
//     before; already 1, nothing; else it stores 1.  All writers store the same value: a benign race.  (A bitmap with one
//     atomicOr per tile and lane -- lane k walking 8 consecutive steps -- was built first: 354 us against 110, every lane's
//     accesses land in a different line.  The end-cell flag first lived in the row-major bitmap of the keyed path: up to 64
//     lines per access for a y-major beam, and four more words per lane in the apply pass.)
//   update_apply_dense_kernel: one wavefront per 64 x 4-cell block of the box (8 tiles = 256 contiguous bytes of the byte
//     map).  It reads the block's bytes, skips the block when all are zero, applies the reference's rule to the marked
//     cells row by row (coalesced 256-byte rows), and clears what it read -- the block has ONE owner, so there is no race
//     on the marks, and the byte map is all zero again between updates (no generation tag to wrap).  Untouched cells cost
//     1 byte instead of 4 bytes + 1 bit.
// Same cells, same rule, same order-dependent artefacts: the maps stay bit-identical to the reference.
#ifndef HSM_MARK_XCD_CHUNK  // workgroups of consecutive beams per XCD turn (xcd_block); < 0: the hardware's round robin
#define HSM_MARK_XCD_CHUNK 16
#endif

// a wave-uniform kernel argument pinned in SGPRs before a loop (left alone, the compiler re-loads it from the kernarg segment
// -- s_load + s_waitcnt -- inside every conditional block of every iteration)
template <class T>
__device__ __forceinline__ T pinned_sgpr(T v) {
  asm volatile("" : "+s"(v));
  return v;
}

// What bounds this kernel was measured (profiles/r03/README.md, what-if builds): with NO memory operation in the walk it
// still took 50 of its 77 us -- instruction issue, not HBM, L2 atomics or load latency (unrolling the walk for four loads in
// flight, several beams per wavefront, plain stores instead of the atomics: no gain).  So the walk is kept short: the cell
// coordinates advance incrementally (no multiplications: v_mad_u64_u32 / v_mul_lo_u32 are quarter rate), the tile row
// offset is a 24-bit multiply, the kernel arguments sit in SGPRs, the per-beam divisions are float reciprocals with an
// exact fix-up (every operand is below 2^24).
__device__ __forceinline__ unsigned int div_small(unsigned int num, unsigned int den) {  // num, den < 2^24, den > 0: exact
  unsigned int q = (unsigned int)(__builtin_amdgcn_rcpf((float)den) * (float)num);
  // the estimate is within one of the quotient
  int rem = (int)(num - q * den);
  if (rem < 0) {
    --q;
    rem += (int)den;
  }
  if (rem >= (int)den) ++q;
  return q;
}

__host__ __device__ __forceinline__ int mark_dense_blocks(int n) {  // workgroups of 4 wavefronts = 4 beams, a multiple of 8
  return ((n + 3) / 4 + 7) / 8 * 8;
}

// Tried in round 4 and dropped (profiles/r04/update_variants_kernel_us.txt, maps bit-identical in every variant): G = 2 / 4
// beams side by side in one wavefront, 64 / G lanes each (the set-up amortised over G beams): 76.8 / 120 us against 72.1 --
// the beams of a group differ in length and octant, so lanes idle and the loop diverges; without the duplicate suppression
// (HSM_MARK_DEDUP=0): 104 us -- the per-step load only sees marks that have reached this XCD's L2, the predecessor test
// needs no memory at all.
__global__ void __launch_bounds__(256) update_mark_free_dense_kernel(const UpdateBatch B) {
  const UpdateParams& P = B.lv[blockIdx.y];
  const int lane = threadIdx.x & 63;
  // Neighbouring beams cross the same cells for most of their length, and a mark byte read from another XCD's L2 is stale
  // (this kernel's stores stay in the writer's L2 until they are evicted): with the hardware's round robin of workgroups
  // over the XCDs every XCD walks every part of the fan.  Chunks of consecutive workgroups per XCD (the grid's x extent is
  // a multiple of 8, so workgroup b of any level runs on XCD b % 8) keep a sector's lines in ONE L2, where the walk sees
  // its neighbours' marks and skips the stores.
  const int wg = HSM_MARK_XCD_CHUNK >= 0 ? xcd_block((int)blockIdx.x, (int)gridDim.x, HSM_MARK_XCD_CHUNK) : (int)blockIdx.x;
  const int beam = __builtin_amdgcn_readfirstlane(wg * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));  // wave-uniform
  if (beam >= P.n) return;
  // the beam's record and its predecessor's: wave-uniform 16-byte loads (scalar cache), everything derived from them
  // stays in SGPRs
  const uint4* __restrict__ recs = reinterpret_cast<const uint4*>(P.recs);
  const uint4 rb = recs[beam];
  const unsigned int da = rb.x;
  if (da == 0u) return;  // skipped beam (beam_line: invalid)
  if ((unsigned int)lane >= da) return;
  const uint4 rp = beam > 0 ? recs[beam - 1] : make_uint4(0u, 0u, 0u, 0u);
  const unsigned int db = rb.y, q64 = rb.z, r64 = rb.w >> 3, oct = rb.w & 7u;
  const unsigned int key = (P.serial << kBeamBits) | (kBeamMask - (unsigned int)beam);
  // lane k visits steps k, k + 64, ...: quotient / remainder of (e0 + i*db) / da carried forward by the per-64-steps increment
  const bool small = da < (1u << 17);  // (e0 + 63 db, 64 db < 2^24: always, for maps below 131072 cells a side)
  const unsigned int num0 = (da >> 1) + (unsigned int)lane * db;  // e0 = abs_da / 2
  unsigned int q = small ? div_small(num0, da) : num0 / da;
  unsigned int r = num0 - q * da;
  // duplicate suppression against the previous beam (mark_free_block): valid, same octant
#ifndef HSM_MARK_DEDUP
#define HSM_MARK_DEDUP 1
#endif
  const bool dedup = HSM_MARK_DEDUP && rp.x != 0u && (rp.w & 7u) == oct;
  const unsigned int pda = dedup ? rp.x : 0u;  // no step is "also the previous beam's" without dedup
  const unsigned int pden = dedup ? rp.x : 1u, pdb = rp.y;
  const unsigned int pnum0 = (pden >> 1) + (unsigned int)lane * pdb;
  unsigned int pq = dedup ? (pden < (1u << 17) ? div_small(pnum0, pden) : pnum0 / pden) : 0u, pr = dedup ? pnum0 - pq * pden : 0u;
  const unsigned int pq64 = dedup ? rp.z : 0u, pr64 = dedup ? rp.w >> 3 : 0u;
  // the walk in (x, y): i steps along the major axis, q along the minor one; both advance by additions
  const bool x_major = (oct & 1u) != 0u;
  const int sgn_a = (oct & 2u) ? 1 : -1, sgn_b = (oct & 4u) ? 1 : -1;
  const int ax = x_major ? sgn_a : 0, ay = x_major ? 0 : sgn_a, mx = x_major ? 0 : sgn_b, my = x_major ? sgn_b : 0;
  int cx = P.bx + ax * lane + mx * (int)q, cy = P.by + ay * lane + my * (int)q;
  const int dx64 = 64 * ax + mx * (int)q64, dy64 = 64 * ay + my * (int)q64;  // per iteration, before the remainder's carry
  const unsigned int tiles_x = pinned_sgpr((unsigned int)P.lv.kf_tiles_x);
  // (global address space spelled out: a pointer that went through pinned_sgpr's asm is a generic one to the compiler, and the
  // walk's byte accesses became flat_load / flat_store with 64-bit per-lane addresses instead of global_* on an SGPR base)
  typedef __attribute__((address_space(1))) unsigned char gbyte;
  typedef __attribute__((address_space(1))) unsigned int gword;
  gbyte* const marks = (gbyte*)pinned_sgpr(P.lv.free_bytes);
  gword* const keys = (gword*)pinned_sgpr(P.lv.key_free);
#if HSM_MARK_TILE16
  const unsigned int mtiles_x = pinned_sgpr((unsigned int)mark_tiles_x(P.lv.sx));
#endif
  auto key_index = [&]() -> unsigned int {  // of the current cell, in the free-key plane
#if HSM_KEYFREE_TILE
    return ((__umul24((unsigned int)cy >> 2, tiles_x) + ((unsigned int)cx >> 3)) << 5) | (((unsigned int)cy & 3u) << 3) |
           ((unsigned int)cx & 7u);  // == key_free_index(P.lv, cx, cy): rows of tiles and tiles per row are below 2^24
#else
    return key_free_index(P.lv, (unsigned int)cx, (unsigned int)cy);
#endif
  };
  auto cell_index = [&]() -> unsigned int {  // of the current cell, in the mark-byte plane
#if HSM_MARK_TILE16
    return ((__umul24((unsigned int)cy >> 3, mtiles_x) + ((unsigned int)cx >> 4)) << 7) | (((unsigned int)cy & 7u) << 4) |
           ((unsigned int)cx & 15u);  // == mark_index(P.lv, cx, cy)
#else
    return key_index();
#endif
  };
  auto advance = [&]() {  // 64 steps on: the carries of both error accumulators
    q += q64;
    r += r64;
    cx += dx64;
    cy += dy64;
    if (r >= da) {
      r -= da;
      ++q;
      cx += mx;
      cy += my;
    }
    pq += pq64;
    pr += pr64;
    if (pr >= pda && dedup) {
      pr -= pda;
      ++pq;
    }
  };
  auto touch = [&](unsigned int kc, unsigned char m, unsigned int kkey) {  // kkey: the cell's index in the free-key plane
    if (m & kMarkEnd) {
      atomicMax((unsigned int*)&keys[kkey], key);  // a beam ends here: the lowest crossing beam index matters (revert artefact)
    } else if (m == 0) {            // (a stale 0 only repeats the store)
      marks[kc] = kMarkCrossed;
    }
  };
#ifndef HSM_MARK_UNROLL  // 2: two steps (i, i + 64) per iteration with both byte loads in flight before either is acted on
#define HSM_MARK_UNROLL 1
#endif
#if HSM_MARK_UNROLL == 2
  for (unsigned int i = lane; i < da; i += 128) {
    const bool need_a = !(i < pda && pq == q);
    const unsigned int kc_a = cell_index(), kk_a = key_index();
    advance();
    const bool need_b = i + 64 < da && !(i + 64 < pda && pq == q);
    const unsigned int kc_b = cell_index(), kk_b = key_index();
    advance();
    unsigned char m_a = kMarkCrossed, m_b = kMarkCrossed;  // "already marked": nothing to do
    if (need_a) m_a = marks[kc_a];
    if (need_b) m_b = marks[kc_b];
    touch(kc_a, m_a, kk_a);
    touch(kc_b, m_b, kk_b);
  }
#else
#if HSM_MARK_TILE_END
  const gbyte* const tile_end = (const gbyte*)pinned_sgpr(P.lv.free_bytes + mark_tile_end_offset(P.lv.sx, P.lv.sy));
#endif
#if defined(HSM_EXPERIMENTS) && defined(HSM_WHATIF_WALK)
  // TIMING EXPERIMENTS ONLY (wrong maps): what a binned LDS tile rasteriser could gain on the walk (round-4 verdict, item 3).
  //   1: no memory operation at all -- the walk's arithmetic alone;  2: the marks go to LDS bytes (ds_write_b8) instead of HBM
  __shared__ unsigned char lds_marks[16384];
  unsigned int sink = 0u;
  for (unsigned int i = lane; i < da; i += 64) {
    if (!(i < pda && pq == q)) {
      const unsigned int kc = cell_index();
      if (HSM_WHATIF_WALK == 2) lds_marks[kc & 16383u] = kMarkCrossed; else sink ^= kc;
    }
    advance();
  }
  if (sink == 0x9e3779b9u || (HSM_WHATIF_WALK == 2 && lds_marks[lane] == 77)) marks[0] = 1;  // (keeps the loop alive)
  return;
#endif
  for (unsigned int i = lane; i < da; i += 64) {  // abs_da free cells: steps 0 .. abs_da-1
    if (!(i < pda && pq == q)) {
      const unsigned int kc = cell_index();
#if HSM_MARK_TILE_END
      if (tile_end[kc >> 7] == 0) {
        marks[kc] = kMarkCrossed;  // no beam ends in this tile: nothing to look at (an already set mark is stored again)
      } else {
        touch(kc, marks[kc], key_index());
      }
#else
      // one byte load from the line the store goes to (the row-major end-cell bitmap cost a y-major beam 64 lines per access)
      touch(kc, marks[kc], HSM_MARK_TILE16 ? key_index() : kc);
#endif
    }
    advance();
  }
#endif
}

// requires HSM_KEYFREE_TILE (the host checks): the box is widened to 64-column / 4-row boundaries; any map width (the tile
// rows are padded to whole blocks, key_free_tiles_x)
#ifndef HSM_APPLY_NT  // 1: the dense apply pass writes its three planes with non-temporal stores -- 12 bytes per touched cell that
                      // nothing reads again before the next update; kept out of the L2 they leave it to the marks and the log-odds
                      // rows (update 0.198 -> 0.178 ms on configs[4]; non-temporal LOADS of the rows or stores of the cleared marks
                      // lose: 0.205 ms)
#define HSM_APPLY_NT 1
#endif
#ifndef HSM_APPLY_BLOCKS  // 64 x 4-cell blocks a wavefront of the dense apply pass has in flight
#define HSM_APPLY_BLOCKS 1
#endif
#if !HSM_MARK_TILE16
template <bool SCATTER_TEXELS>
__global__ void __launch_bounds__(256) update_apply_dense_kernel(const UpdateBatch B) {
  const UpdateParams& P = B.lv[blockIdx.y];
  if (P.x1 < P.x0) return;  // this level has nothing to apply
  const int lane = threadIdx.x & 63;
  const int bx0 = P.x0 & ~63, by0 = P.y0 & ~3;
  const int nbx = ((P.x1 | 63) - bx0 + 1) >> 6, nby = (((P.y1 | 3) - by0) >> 2) + 1;
  const int nblocks = nbx * nby;
  const int waves = (int)((gridDim.x * blockDim.x) >> 6);
  const int sx = P.lv.sx, sy = P.lv.sy;
  // A wavefront's blocks are a chain of memory round trips (marks -> log-odds rows -> stores), a dozen blocks long, and the
  // pass was waiting for them 84 % of the time (profiles/r03/README.md).  So the marks of the NEXT block are requested
  // before this one is processed, and the rows of a block are all requested before the first is computed.
  // the block's 256 mark bytes: lane l reads ONE dword of them (tile l / 8, row (l % 8) / 2, half l % 2)
  auto marks_of = [&](int blk) -> unsigned int* {
    const int X0 = bx0 + ((blk % nbx) << 6), Y0 = by0 + ((blk / nbx) << 2);
    const unsigned int tile0 = (((unsigned int)(Y0 >> 2) * (unsigned int)P.lv.kf_tiles_x) + (unsigned int)(X0 >> 3)) << 5;  // byte index
    return reinterpret_cast<unsigned int*>(P.lv.free_bytes + tile0) + lane;
  };
  // kApplyBlocks blocks per iteration: their rows are all in flight together (Little's law: 32 wavefronts per CU with
  // four 256-byte rows each in flight sustain ~4 TB/s at this latency, which is what the one-block form measured)
  constexpr int NBLK = HSM_APPLY_BLOCKS;
  int blk = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * NBLK;
  if (blk >= nblocks) return;
  unsigned int fw_next[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) fw_next[j] = blk + j < nblocks ? *marks_of(blk + j) : 0u;
  for (; blk < nblocks; blk += waves * NBLK) {
    unsigned int fw[NBLK];
    bool live[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) fw[j] = fw_next[j];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) fw_next[j] = blk + waves * NBLK + j < nblocks ? *marks_of(blk + waves * NBLK + j) : 0u;
    bool fre[NBLK][4], occ[NBLK][4];
    float l[NBLK][4];
    unsigned int ko[NBLK][4], kf[NBLK][4];
    // phase 1: classify the four cells of this lane's column in every block and request what their update needs
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
      live[j] = blk + j < nblocks && __ballot(fw[j] != 0u) != 0ull;  // wave-uniform: something of this scan in the block
      if (!live[j]) continue;
      const int X0 = bx0 + (((blk + j) % nbx) << 6), Y0 = by0 + (((blk + j) / nbx) << 2);
      const int x = X0 + lane;
      if (fw[j] != 0u) *marks_of(blk + j) = 0u;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        const int y = Y0 + dy;
        const size_t c = (size_t)y * sx + x;
        // the byte of cell (x, Y0 + dy): tile lane / 8, byte dy * 8 + lane % 8 = dword (lane & ~7) + 2 dy + (lane & 7) / 4, byte lane & 3
        const unsigned int fwd = (unsigned int)__shfl((int)fw[j], (lane & ~7) + 2 * dy + ((lane & 7) >> 2));
        const unsigned int mark = (fwd >> ((lane & 3) << 3)) & 0xffu;
        occ[j][dy] = (mark & kMarkEnd) != 0u && y < sy && x < sx;  // (the widened box may reach past the map's last row / column:
        fre[j][dy] = (mark & kMarkCrossed) != 0u && y < sy && x < sx;  //  no mark can sit there, cheap to insist)
        l[j][dy] = 0.0f;
        ko[j][dy] = kf[j][dy] = 0u;
        if (fre[j][dy] || occ[j][dy]) l[j][dy] = P.lv.logodds[c];
        if (occ[j][dy]) {
          ko[j][dy] = P.lv.key_occ[c];
          kf[j][dy] = P.lv.key_free[key_free_index(P.lv, (unsigned int)x, (unsigned int)y)];
        }
      }
    }
    // phase 2: the reference's rule, row by row (coalesced 256-byte rows)
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
      if (!live[j]) continue;
      const int X0 = bx0 + (((blk + j) % nbx) << 6), Y0 = by0 + (((blk + j) / nbx) << 2);
      const int x = X0 + lane;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        const int y = Y0 + dy;
        const size_t c = (size_t)y * sx + x;
        bool is_occ = occ[j][dy], is_fre = fre[j][dy];
        if (is_occ) {
          is_occ = (ko[j][dy] >> kBeamBits) == P.serial;  // (a flag without this scan's key cannot occur; cheap to insist)
          is_fre = (kf[j][dy] >> kBeamBits) == P.serial;
        }
        if (!is_fre && !is_occ) continue;
        float lo = l[j][dy];
        int stamp;
        if (is_occ) {
          // free-touched by an earlier beam of this scan: applied, then reverted (OccGridMapBase.h:231-233)
          if (is_fre && (kBeamMask - (kf[j][dy] & kBeamMask)) < (kBeamMask - (ko[j][dy] & kBeamMask))) {
            lo += P.log_odds_free;
            lo -= P.log_odds_free;
          }
          if (lo < 50.0f) lo += P.log_odds_occ;  // updateSetOccupied
          stamp = P.mark_occ;
        } else {
          lo += P.log_odds_free;                 // updateSetFree
          stamp = P.mark_free;
        }
#if HSM_APPLY_NT
        __builtin_nontemporal_store(lo, &P.lv.logodds[c]);
        __builtin_nontemporal_store(stamp, &P.lv.update_index[c]);
        const float p = grid_probability(lo);
        __builtin_nontemporal_store(p, &P.lv.prob[c]);
#else
        P.lv.logodds[c] = lo;
        P.lv.update_index[c] = stamp;
        const float p = grid_probability(lo);
        P.lv.prob[c] = p;
#endif
        if (SCATTER_TEXELS) {
          float* q = reinterpret_cast<float*>(P.lv.quad);
          const bool lastx = x == sx - 1, lasty = y == sy - 1;
          auto put = [&](int tx, int ty, int comp) { q[4 * (size_t)quad_index(tx, ty, P.lv.tiles_x, sx) + comp] = p; };
          put(x, y, 0);
          if (x > 0) put(x - 1, y, 1);
          if (lastx) put(x, y, 1);
          if (y > 0) put(x, y - 1, 2);
          if (lasty) put(x, y, 2);
          if (x > 0 && y > 0) put(x - 1, y - 1, 3);
          if (lastx && y > 0) put(x, y - 1, 3);
          if (lasty && x > 0) put(x - 1, y, 3);
          if (lastx && lasty) put(x, y, 3);
        }
      }
    }
  }
}

#endif  // !HSM_MARK_TILE16

#if HSM_MARK_TILE16
// HSM_MARK_TILE16: the dense apply pass on 32 x 8-cell blocks (two 16 x 8 mark tiles = 256 contiguous mark bytes, one dword
// per lane).  Lane l works on column l % 32 of the block and on rows 2 i + l / 32 (i = 0 .. 3): every access of the log-odds /
// stamp / probability planes is two 128-byte row segments per wavefront.  Otherwise update_apply_dense_kernel's block: skip
// when no mark is set, the reference's rule on the marked cells, marks cleared; the next block's marks are requested first.
template <bool SCATTER_TEXELS>
__global__ void __launch_bounds__(256) update_apply_dense_kernel(const UpdateBatch B) {
  const UpdateParams& P = B.lv[blockIdx.y];
  if (P.x1 < P.x0) return;
  const int lane = threadIdx.x & 63;
  const int xl = lane & 31, rh = lane >> 5;
  const int bx0 = P.x0 & ~31, by0 = P.y0 & ~7;
  const int nbx = ((P.x1 | 31) - bx0 + 1) >> 5, nby = (((P.y1 | 7) - by0) >> 3) + 1;
  const int nblocks = nbx * nby;
  const int waves = (int)((gridDim.x * blockDim.x) >> 6);
  const int sx = P.lv.sx, sy = P.lv.sy;
  const unsigned int mtx = (unsigned int)mark_tiles_x(sx);
  auto marks_of = [&](int blk) -> unsigned int* {
    const int X0 = bx0 + ((blk % nbx) << 5), Y0 = by0 + ((blk / nbx) << 3);
    const unsigned int t0 = (((unsigned int)(Y0 >> 3) * mtx) + (unsigned int)(X0 >> 4)) << 7;  // byte index of the block's first tile
    return reinterpret_cast<unsigned int*>(P.lv.free_bytes + t0) + lane;
  };
  int blk = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (blk >= nblocks) return;
  unsigned int fw_next = *marks_of(blk);
  for (; blk < nblocks; blk += waves) {
    const unsigned int fw = fw_next;
    fw_next = blk + waves < nblocks ? *marks_of(blk + waves) : 0u;
    if (__ballot(fw != 0u) == 0ull) continue;  // wave-uniform: nothing of this scan in the block
    const int X0 = bx0 + ((blk % nbx) << 5), Y0 = by0 + ((blk / nbx) << 3);
    const int x = X0 + xl;
    if (fw != 0u) *marks_of(blk) = 0u;
#if HSM_MARK_TILE_END
    if (lane < 2) {
      const unsigned int t0 = (((unsigned int)(Y0 >> 3) * mtx) + (unsigned int)(X0 >> 4));
      P.lv.free_bytes[mark_tile_end_offset(sx, sy) + t0 + lane] = 0;
    }
#endif
    bool fre[4], occ[4];
    float l[4];
    unsigned int ko[4], kf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 2 * i + rh, y = Y0 + r;
      const size_t c = (size_t)y * sx + x;
      // the byte of cell (xl, r): tile xl / 16, byte r * 16 + xl % 16 = dword (xl / 16) * 32 + r * 4 + (xl % 16) / 4, byte xl & 3
      const unsigned int fwd = (unsigned int)__shfl((int)fw, ((xl >> 4) << 5) + (r << 2) + ((xl & 15) >> 2));
      const unsigned int mark = (fwd >> ((xl & 3) << 3)) & 0xffu;
      occ[i] = (mark & kMarkEnd) != 0u && y < sy && x < sx;
      fre[i] = (mark & kMarkCrossed) != 0u && y < sy && x < sx;
      l[i] = 0.0f;
      ko[i] = kf[i] = 0u;
      if (fre[i] || occ[i]) l[i] = P.lv.logodds[c];
      if (occ[i]) {
        ko[i] = P.lv.key_occ[c];
        kf[i] = P.lv.key_free[key_free_index(P.lv, (unsigned int)x, (unsigned int)y)];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = Y0 + 2 * i + rh;
      const size_t c = (size_t)y * sx + x;
      bool is_occ = occ[i], is_fre = fre[i];
      if (is_occ) {
        is_occ = (ko[i] >> kBeamBits) == P.serial;
        is_fre = (kf[i] >> kBeamBits) == P.serial;
      }
      if (!is_fre && !is_occ) continue;
      float lo = l[i];
      int stamp;
      if (is_occ) {
        if (is_fre && (kBeamMask - (kf[i] & kBeamMask)) < (kBeamMask - (ko[i] & kBeamMask))) {
          lo += P.log_odds_free;
          lo -= P.log_odds_free;
        }
        if (lo < 50.0f) lo += P.log_odds_occ;
        stamp = P.mark_occ;
      } else {
        lo += P.log_odds_free;
        stamp = P.mark_free;
      }
      __builtin_nontemporal_store(lo, &P.lv.logodds[c]);
      __builtin_nontemporal_store(stamp, &P.lv.update_index[c]);
      const float p = grid_probability(lo);
      __builtin_nontemporal_store(p, &P.lv.prob[c]);
      if (SCATTER_TEXELS) {
        float* q = reinterpret_cast<float*>(P.lv.quad);
        const bool lastx = x == sx - 1, lasty = y == sy - 1;
        auto put = [&](int tx, int ty, int comp) { q[4 * (size_t)quad_index(tx, ty, P.lv.tiles_x, sx) + comp] = p; };
        put(x, y, 0);
        if (x > 0) put(x - 1, y, 1);
        if (lastx) put(x, y, 1);
        if (y > 0) put(x, y - 1, 2);
        if (lasty) put(x, y, 2);
        if (x > 0 && y > 0) put(x - 1, y - 1, 3);
        if (lastx && y > 0) put(x, y - 1, 3);
        if (lasty && x > 0) put(x - 1, y, 3);
        if (lastx && lasty) put(x, y, 3);
      }
    }
  }
}
#endif

// dense over the box grown by one cell towards -x/-y: texel (x,y) holds P of (x..x+1, y..y+1)
__global__ void __launch_bounds__(256) update_texels_kernel(const UpdateBatch B) {
  const UpdateParams& P = B.lv[blockIdx.y];
  if (P.x1 < P.x0) return;
  const int tx0 = P.x0 > 0 ? P.x0 - 1 : 0, ty0 = P.y0 > 0 ? P.y0 - 1 : 0;
  const int w = P.x1 - tx0 + 1, h = P.y1 - ty0 + 1;
  const size_t n = (size_t)w * h;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int x = tx0 + (int)(t % (size_t)w), y = ty0 + (int)(t / (size_t)w);
    const int xn = x + 1 < P.lv.sx ? x + 1 : x;
    const int yn = y + 1 < P.lv.sy ? y + 1 : y;
    const float* p = P.lv.prob;
    P.lv.quad[quad_index(x, y, P.lv.tiles_x, P.lv.sx)] =
        make_float4(p[(size_t)y * P.lv.sx + x], p[(size_t)y * P.lv.sx + xn], p[(size_t)yn * P.lv.sx + x],
                    p[(size_t)yn * P.lv.sx + xn]);
  }
}

// test hook (hsm_debug_marks_nonzero): the dense update's byte map and the keyed update's end-cell bitmap must be ALL ZERO
// between updates (each apply pass clears what its mark passes set; map_update.h "dense scans"): count the non-zero words
__global__ void __launch_bounds__(256) count_nonzero_words_kernel(const unsigned int* __restrict__ words, size_t n,
                                                                  unsigned long long* __restrict__ out) {
  unsigned int local = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    local += words[i] != 0u ? 1u : 0u;
  const unsigned long long m = __ballot(local != 0u);
  if (m == 0ull) return;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) local += (unsigned int)__shfl_down((int)local, d);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)local);
}

// ---- whole-plane maintenance (create / reset / upload) ------------------------------
__global__ void fill_level_kernel(LevelRW L, float logodds, int update_index) {
  const size_t n = (size_t)L.sx * L.sy;
  const float p = grid_probability(logodds);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    L.logodds[i] = logodds;
    L.update_index[i] = update_index;
    L.prob[i] = p;
  }
  // every texel of the tiled plane, including the padding of partial edge tiles
  if (L.quad) {  // the plane layout keeps no texel plane
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)L.quad_texels;
         i += (size_t)gridDim.x * blockDim.x) {
      L.quad[i] = make_float4(p, p, p, p);
    }
  }
}

// rectangle (x0,y0,w,h) of the two SoA planes -> the reference's AoS LogOddsCell {float, int}
__global__ void pack_cells_kernel(LevelRW L, int x0, int y0, int w, int h, int2* __restrict__ out) {
  const size_t n = (size_t)w * h;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = x0 + (int)(i % (size_t)w), y = y0 + (int)(i / (size_t)w);
    const size_t c = (size_t)y * L.sx + x;
    out[i] = make_int2(__float_as_int(L.logodds[c]), L.update_index[c]);
  }
}

// ---- rows next to the path ----------------------------------------------------
// publishMap's cell loop (hector_mapping/src/HectorMappingRos.cpp:449-468): four cells per thread,
// one coalesced 16-byte load and one 4-byte store
__global__ void occupancy_grid_kernel(const float* __restrict__ logodds, size_t n, signed char* __restrict__ out) {
  const size_t n4 = n / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 l = reinterpret_cast<const float4*>(logodds)[i];
    char4 o;
    o.x = l.x < 0.0f ? 0 : (l.x > 0.0f ? 100 : -1);  // isFree / isOccupied, GridMapLogOdds.h:76-84
    o.y = l.y < 0.0f ? 0 : (l.y > 0.0f ? 100 : -1);
    o.z = l.z < 0.0f ? 0 : (l.z > 0.0f ? 100 : -1);
    o.w = l.w < 0.0f ? 0 : (l.w > 0.0f ? 100 : -1);
    reinterpret_cast<char4*>(out)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float l = logodds[n4 * 4 + threadIdx.x];
    out[n4 * 4 + threadIdx.x] = l < 0.0f ? 0 : (l > 0.0f ? 100 : -1);
  }
}

// f4: DistanceMeasurementProvider::getDist (hector_map_tools/.../HectorMapTools.h:133-234) for a batch of
// rays, straight on the log-odds plane (a cell of the published grid is 100 <=> logOdds > 0).  One
// wavefront per ray: lane k tests Bresenham steps k, k+64, ... through the closed form of the error
// accumulator, a ballot finds the FIRST occupied step, and the search stops at that chunk -- the
// sequential early-exit walk of the reference without walking sequentially.
struct RayQueryParams {
  const float* logodds;
  int sx, sy;
  float origin_x, origin_y, scale, inv_scale;  // CoordinateTransformer (:58-98)
  const float2* begin_world;
  const float2* end_world;
  int n;
  float* out_dist;
  float2* out_hit;
};

__global__ void __launch_bounds__(256) ray_distance_kernel(const RayQueryParams P) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= P.n) return;
  const float2 bw = P.begin_world[r], ew = P.end_world[r];
  // getC2Coords(...).cast<int>(): ((world - origo) * inv_scale), truncated
  const int x0 = (int)((bw.x - P.origin_x) * P.inv_scale), y0 = (int)((bw.y - P.origin_y) * P.inv_scale);
  const int x1 = (int)((ew.x - P.origin_x) * P.inv_scale), y1 = (int)((ew.y - P.origin_y) * P.inv_scale);
  float dist = -1.0f;
  if (!((x0 < 0) || (x0 >= P.sx) || (y0 < 0) || (y0 >= P.sy)) && !((x1 < 0) || (x1 >= P.sx) || (y1 < 0) || (y1 >= P.sy))) {
    const int dx = x1 - x0, dy = y1 - y0;
    const unsigned int abs_dx = (unsigned int)(dx < 0 ? -dx : dx), abs_dy = (unsigned int)(dy < 0 ? -dy : dy);
    const int offset_dx = dx > 0 ? 1 : -1;
    const int offset_dy = (dy > 0 ? 1 : -1) * P.sx;
    BeamLine b;
    b.start = (unsigned int)(y0 * P.sx + x0);
    if (abs_dx >= abs_dy) {
      b.abs_da = abs_dx; b.abs_db = abs_dy; b.offset_a = offset_dx; b.offset_b = offset_dy;
    } else {
      b.abs_da = abs_dy; b.abs_db = abs_dx; b.offset_a = offset_dy; b.offset_b = offset_dx;
    }
    b.e0 = b.abs_da / 2;
    const unsigned int end = b.abs_da < 5000u ? b.abs_da : 5000u;  // bresenham2D(..., max_length = 5000)
    for (unsigned int i0 = 0; i0 < end; i0 += 64) {
      const unsigned int i = i0 + lane;
      const bool occ = i < end && P.logodds[line_cell(b, i)] > 0.0f;  // data[offset] == 100
      const unsigned long long m = __ballot(occ);
      if (m) {
        const unsigned int ih = i0 + (unsigned int)__ffsll((long long)m) - 1u;
        const unsigned int c = line_cell(b, ih);
        const int ex = (int)(c % (unsigned int)P.sx), ey = (int)(c / (unsigned int)P.sx);
        const float fx = (float)(x0 - ex), fy = (float)(y0 - ey);
        dist = (float)(int)sqrtf(fx * fx + fy * fy);  // int distMap = (begin - end).cast<float>().norm()
        if (lane == 0) P.out_hit[r] = make_float2(P.origin_x + ((float)ex * P.scale), P.origin_y + ((float)ey * P.scale));
        break;
      }
    }
  }
  if (lane == 0) P.out_dist[r] = P.scale * dist;  // getC1Scale
}

// rosLaserScanToDataContainer (HectorMappingRos.cpp:483-507).  trig[i] = (cosf(angle_i), sinf(angle_i))
// comes from the host (the running fp32 angle and the libm calls are the node's); this kernel applies
// the range gate, the scale and the products, and compacts the survivors IN ORDER: one workgroup,
// chunks of 1024 beams, wave ballot + LDS for the ordered offsets.
__global__ void __launch_bounds__(1024) ingest_laser_scan_kernel(const float* __restrict__ ranges,
                                                                const float2* __restrict__ trig, int n,
                                                                float range_min, float max_range_for_container,
                                                                float scale_to_map, float2* __restrict__ out,
                                                                int* __restrict__ out_n) {
  __shared__ int wave_count[16];
  __shared__ int base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    float dist = i < n ? ranges[i] : 0.0f;
    const bool keep = (i < n) && (dist > range_min) && (dist < max_range_for_container);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_count[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_count[w];
    if (keep) {
      dist *= scale_to_map;
      const float2 cs = trig[i];
      out[off + __popcll(m & ((1ull << lane) - 1ull))] = make_float2(cs.x * dist, cs.y * dist);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wave_count[w];
      base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_n = base;
}

// rosPointCloudToDataContainer (HectorMappingRos.cpp:509-542), optionally preceded by
// laser_geometry's projectLaser (the node's default path, :273-282; third party, algorithm stated in
// include/hector_mi355/capi.h): one pass, ordered compaction like the kernel above.  tf arithmetic is fp64
// (tfScalar), the gates and the products fp32, exactly the node's types.
struct CloudIngestParams {
  const float* pts_xyz;  // [n,3] geometry_msgs::Point32, or nullptr when projecting from ranges
  const float* ranges;   // projectLaser input, or nullptr
  const double2* unit;   // (cos, sin)(angle_min + (double)i * angle_increment): sensor constants from the host
  int n;
  float range_min;       // projectLaser gate: range < range_cutoff (double compare) && range >= range_min
  double range_cutoff;
  double T[12];          // laser -> base transform, rows [R | t]
  float sqr_min, sqr_max, z_min, z_max, scale;
  float2* out;
  int* out_n;
};

__global__ void __launch_bounds__(1024) ingest_point_cloud_kernel(CloudIngestParams P) {
  __shared__ int wave_count[16];
  __shared__ int base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < P.n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    bool keep = false;
    float2 e = make_float2(0.0f, 0.0f);
    if (i < P.n) {
      float px, py, pz;
      bool valid = true;
      if (P.ranges) {
        const float range = P.ranges[i];
        const double r = (double)range;
        const double2 u = P.unit[i];
        px = (float)(r * u.x);
        py = (float)(r * u.y);
        pz = 0.0f;
        valid = ((double)range < P.range_cutoff) && (range >= P.range_min);
      } else {
        px = P.pts_xyz[3 * (size_t)i];
        py = P.pts_xyz[3 * (size_t)i + 1];
        pz = P.pts_xyz[3 * (size_t)i + 2];
      }
      const float dist_sqr = px * px + py * py;
      if (valid && (dist_sqr > P.sqr_min) && (dist_sqr < P.sqr_max) && !((px < 0.0f) && (dist_sqr < 0.50f))) {
        const double vx = px, vy = py, vz = pz;
        const double bx = (P.T[0] * vx + P.T[1] * vy + P.T[2] * vz) + P.T[3];
        const double by = (P.T[4] * vx + P.T[5] * vy + P.T[6] * vz) + P.T[7];
        const double bz = (P.T[8] * vx + P.T[9] * vy + P.T[10] * vz) + P.T[11];
        const float zl = (float)(bz - P.T[11]);
        if (zl > P.z_min && zl < P.z_max) {
          keep = true;
          e = make_float2((float)bx * P.scale, (float)by * P.scale);
        }
      }
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_count[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_count[w];
    if (keep) P.out[off + __popcll(m & ((1ull << lane) - 1ull))] = e;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wave_count[w];
      base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *P.out_n = base;
}

// device expf / getGridProbability sweep for the parity tests
__global__ void expf_debug_kernel(const float* __restrict__ x, int n, float* __restrict__ e, float* __restrict__ p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  e[i] = libm::expf_glibc(x[i]);
  p[i] = grid_probability(x[i]);
}

__global__ void rebuild_prob_kernel(LevelRW L) {
  const size_t n = (size_t)L.sx * L.sy;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    L.prob[i] = grid_probability(L.logodds[i]);
  }
}

// texel (x,y) from the probability plane; the last row/column (never sampled: the
// bounds test keeps ix,iy <= size-2) replicate the edge
__global__ void rebuild_quad_kernel(LevelRW L) {
  const size_t n = (size_t)L.sx * L.sy;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % (size_t)L.sx), y = (int)(i / (size_t)L.sx);
    const int x1 = x + 1 < L.sx ? x + 1 : x;
    const int y1 = y + 1 < L.sy ? y + 1 : y;
    L.quad[quad_index(x, y, L.tiles_x, L.sx)] = make_float4(L.prob[(size_t)y * L.sx + x], L.prob[(size_t)y * L.sx + x1],
                            L.prob[(size_t)y1 * L.sx + x], L.prob[(size_t)y1 * L.sx + x1]);
  }
}

}  // namespace hsm
