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
//   pass 1 "mark":  one wavefront per beam; lane k owns Bresenham steps k, k+64, ...
//       The cell of step i has a closed form (minor steps = floor((e0 + i*db)/da)), so
//       no lane walks the line sequentially.  Each touched cell receives
//       atomicMax(key) with key = (scan serial << 16) | (0xFFFF - beam index) on the
//       free-key plane (line cells) or the occ-key plane (end cell): the plane then
//       holds, per cell, the FIRST beam (lowest index) that touched it in this scan.
//   pass 2 "apply": same geometry; the unique lane whose key won a cell applies the
//       update to the log-odds plane, writes the reference's updateIndex stamp and
//       refreshes the probability plane and the four quad texels that contain the
//       cell.  No float atomics, no races: every word has exactly one writer.
// Keys of earlier scans are always smaller than the current ones, so the key planes
// never need clearing (only when the 16-bit serial wraps, every 65535 updates).
//
// Traffic: per touched cell 2 key atomics + 8 B log-odds/stamp RMW +
// 20 B probability/texel stores; HBM-bound scattered integer/byte work, no MFMA.
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
  int sx, sy;
  int tiles_x, quad_texels;  // tiled texel plane geometry (gn_match.h quad_index)
};

struct UpdateParams {
  LevelRW lv;
  Affine2 pose;           // Translation(mapPose.xy) * Rotation(mapPose.theta), host sinf/cosf
  const float2* pts;      // level-0 endpoints (robot frame)
  int n;
  float pt_scale;         // 2^-level (exact), DataPointContainer.h:46-58
  int bx, by;             // scanBeginMapi (OccGridMapBase.h:137)
  unsigned int serial;    // 1..65535
  float log_odds_free, log_odds_occ;
  int mark_free, mark_occ;  // currMarkFreeIndex / currMarkOccIndex (OccGridMapBase.h:123-124)
};

// GridMapLogOddsFunctions::getGridProbability (GridMapLogOdds.h:163-166);
// expf evaluated in fp64 and rounded once.
__device__ __forceinline__ float grid_probability(float log_odds) {
  const float odds = (float)exp((double)log_odds);
  return odds / (odds + 1.0f);
}

// write P of cell (x,y) into the probability plane and the 4 texels that contain it
__device__ __forceinline__ void store_probability(const LevelRW& L, int x, int y, float p) {
  const int idx = y * L.sx + x;
  L.prob[idx] = p;
  float* q = reinterpret_cast<float*>(L.quad);
  q[4 * quad_index(x, y, L.tiles_x, L.sx) + 0] = p;                              // texel (x,   y  ).P00
  if (x > 0) q[4 * quad_index(x - 1, y, L.tiles_x, L.sx) + 1] = p;               // texel (x-1, y  ).P10
  if (y > 0) q[4 * quad_index(x, y - 1, L.tiles_x, L.sx) + 2] = p;               // texel (x,   y-1).P01
  if (x > 0 && y > 0) q[4 * quad_index(x - 1, y - 1, L.tiles_x, L.sx) + 3] = p;  // texel (x-1, y-1).P11
}

struct BeamLine {
  bool valid;
  int x1, y1;
  unsigned int abs_da, abs_db;
  int offset_a, offset_b;
  unsigned int e0;
  unsigned int start;
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

__global__ void __launch_bounds__(256) update_mark_kernel(const UpdateParams P) {
  const int lane = threadIdx.x & 63;
  const int beam = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (beam >= P.n) return;
  const BeamLine b = beam_line(P, beam);
  if (!b.valid) return;
  const unsigned int key = (P.serial << 16) | (0xFFFFu - (unsigned int)beam);
  for (unsigned int i = lane; i < b.abs_da; i += 64) {  // abs_da free cells: steps 0 .. abs_da-1
    atomicMax(&P.lv.key_free[line_cell(b, i)], key);
  }
  if (lane == 0) {
    atomicMax(&P.lv.key_occ[(unsigned int)(b.y1 * P.lv.sx + b.x1)], key);
  }
}

__global__ void __launch_bounds__(256) update_apply_kernel(const UpdateParams P) {
  const int lane = threadIdx.x & 63;
  const int beam = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (beam >= P.n) return;
  const BeamLine b = beam_line(P, beam);
  if (!b.valid) return;
  const unsigned int key = (P.serial << 16) | (0xFFFFu - (unsigned int)beam);
  for (unsigned int i = lane; i < b.abs_da; i += 64) {
    const unsigned int c = line_cell(b, i);
    if (P.lv.key_free[c] != key) continue;                  // another beam touched it first
    if ((P.lv.key_occ[c] >> 16) == P.serial) continue;      // occupied wins, its owner handles it
    float l = P.lv.logodds[c];
    l += P.log_odds_free;                                   // updateSetFree
    P.lv.logodds[c] = l;
    P.lv.update_index[c] = P.mark_free;
    store_probability(P.lv, (int)(c % (unsigned int)P.lv.sx), (int)(c / (unsigned int)P.lv.sx),
                      grid_probability(l));
  }
  if (lane == 0) {
    const unsigned int c = (unsigned int)(b.y1 * P.lv.sx + b.x1);
    if (P.lv.key_occ[c] == key) {
      float l = P.lv.logodds[c];
      const unsigned int kf = P.lv.key_free[c];
      if ((kf >> 16) == P.serial && (0xFFFFu - (kf & 0xFFFFu)) < (unsigned int)beam) {
        // free-touched by an earlier beam of this scan: applied, then reverted (:231-233)
        l += P.log_odds_free;
        l -= P.log_odds_free;
      }
      if (l < 50.0f) l += P.log_odds_occ;                   // updateSetOccupied
      P.lv.logodds[c] = l;
      P.lv.update_index[c] = P.mark_occ;
      store_probability(P.lv, b.x1, b.y1, grid_probability(l));
    }
  }
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
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)L.quad_texels;
       i += (size_t)gridDim.x * blockDim.x) {
    L.quad[i] = make_float4(p, p, p, p);
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
