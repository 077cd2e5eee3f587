// gn_match.h -- the scan-to-map Gauss-Newton matcher as CDNA4 (gfx950) HIP kernels.
//
// What it computes (reference, HSL/ = hector_mapping/include/hector_slam_lib/):
//   per beam   OccGridMapUtil::interpMapValueWithDerivatives   HSL/map/OccGridMapUtil.h:287-347
//   per scan   OccGridMapUtil::getCompleteHessianDerivs        HSL/map/OccGridMapUtil.h:64-104
//   per step   ScanMatcher::estimateTransformationLogLh        HSL/matcher/ScanMatcher.h:194-221
//   per level  ScanMatcher::matchData                          HSL/matcher/ScanMatcher.h:54-190
//   per match  MapRepMultiMap::matchData                       HSL/slam_main/MapRepMultiMap.h:116-132
//
// How it is mapped onto the machine:
//   * one TEAM of WPS wavefronts (64 lanes each) owns one (pose hypothesis, scan)
//     pair for the WHOLE coarse-to-fine schedule (all levels, all GN steps) -- the
//     14-step dependent chain never leaves the CU, no host round trips;
//   * beams are dealt round-robin to lanes (beam i -> lane i mod 64*WPS), so the
//     float2 endpoint loads of a wavefront are one contiguous 512-byte segment and
//     neighbouring lanes sample neighbouring map cells;
//   * the occupancy pyramid is sampled from a texture-like "quad" plane: texel
//     (x,y) = float4{P(x,y), P(x+1,y), P(x,y+1), P(x+1,y+1)} -> ONE 16-byte gather
//     per beam instead of four 4-byte gathers on two rows (HSM_LAYOUT_PLANE samples
//     the probability plane directly and keeps no texel plane: less memory, one pass
//     less per map update);
//   * the 6 unique H terms + 3 dTr terms are lane-local fp32 partial sums, reduced
//     with ONE folded wavefront reduction (v_permlane32/16_swap on pairs of values,
//     DPP bank masks on the row levels, 9 v_readlane: 34 instructions, no LDS),
//     then -- when WPS > 1 -- staged through LDS (double buffered, one barrier per
//     GN step);
//   * every lane ends up with bit-identical totals and solves the 3x3 system
//     redundantly: no broadcast, no divergence;
//   * batches of long scans run the texel-cache form (gn_match_cached_kernel): the
//     last texel of every beam stays in VGPRs, endpoints in LDS, gathers only in the
//     lanes whose cell changed, issued one beam ahead with counted s_waitcnt (inline
//     asm), first GN step peeled so that it runs while the endpoints stream in;
//   * a single DENSE scan (>= 4096 beams): in the default mode one workgroup whose first wavefront
//     adds while fifteen produce one round ahead of it (gn_match_exact_dense_kernel); with
//     HSM_PARITY_FAST up to 64 workgroups that exchange tagged partial sums (gn_match_coop_kernel).
//   No MFMA: this is a bilinear gather plus a 9-term reduction, not a contraction.
//   Measured limits (profiles/r02/README.md): VALU issue (61 instructions per beam) and
//   the texture path of the divergent 16-byte gathers -- not HBM.
//
// Numerics: built with -ffp-contract=off.  Every per-beam value (M, dM/dx, dM/dy,
// rotDeriv and the nine products) is the same IEEE fp32 expression, in the same
// order, as the reference.  In the DEFAULT mode (HSM_PARITY_AUTO, and HSM_PARITY_EXACT) the
// nine per-beam products are staged through LDS and summed by nine lanes in beam order,
// i = 0 .. n-1, exactly the reference's fp32 chains: identical bits on every entry point.
// The opt-in HSM_PARITY_FAST changes only the ORDER of the beam summation (strided partial
// sums + tree instead of one sequential chain).  sinf/cosf/expf are glibc's algorithms
// operation for operation (libm_exact.h): identical bits.
#pragma once
// Measured variants that are not shipped (the two-wave texel-cache form, per-wave time stamps) only compile with
// -DHSM_EXPERIMENTS; the default library holds the shipped forms alone.
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>

#include "libm_exact.h"
#include "pose_exchange.h"

namespace hsm {

constexpr int kMaxLevels = 8;
constexpr int kLayoutQuad = 1;
constexpr int kLayoutPlane = 2;
// HSM_PIPELINE=1 issues the gather of beam k+1 before beam k is consumed; HSM_UNROLL=n gathers n beams
// back to back before consuming them.  Measured on MI355X (profiles/r01/README.md): one beam at a
// time is fastest (372 M it/s; pipelined 358; chunks of 2 / 4 / 6 / 9: 359 / 342 / 340 / 309) --
// the four waves per SIMD already interleave, extra texels in flight only cost registers.
#ifndef HSM_PIPELINE
#define HSM_PIPELINE 0
#endif
#ifndef HSM_UNROLL
#define HSM_UNROLL 1
#endif
constexpr int kUnroll = HSM_UNROLL;  // texel gathers a lane keeps in flight (register-resident form)

// Eigen::Affine2f as the reference builds it: 2x2 linear (column major) + translation.
struct Affine2 {
  float l00, l10, l01, l11, t0, t1;
};

// read-only view of one pyramid level for the matcher
struct LevelView {
  const float4* quad;  // texels {P00,P10,P01,P11}, tiled (quad_index), + one all-zero texel at quad_texels
  const float* prob;   // [sy*sx] plain probability plane (row major) + sx+2 zero cells
  int sx, sy;
  int tiles_x;         // quad tiles per row = ceil(sx / 4)
  int quad_texels;     // tiles_x * ceil(sy / 2) * 8
  float limx, limy;    // dims - 2  (MapDimensionProperties.h:70-74)
  Affine2 mapTworld;   // GridMapBase.h:272
  Affine2 worldTmap;   // GridMapBase.h:279
  float pt_scale;      // 2^-level applied to the level-0 endpoints (DataPointContainer.h:46-58)
  int gn_steps;        // 1 + maxIterations (ScanMatcher.h:74,94-97)
};

struct SpecStats {  // optional counters of gn_match_spec_kernel: boundaries walked, candidate == carry, shifted, re-run
  unsigned long long boundaries, exact, shifted, rerun;
};

struct MatchParams {
  LevelView lv[kMaxLevels];
  int first_level;         // coarsest level to run (levels first_level .. last_level, descending)
  int last_level;
  int batch;
  const float* begin_world;  // [B*3], or nullptr: the single start estimate travels in begin_inline
  float begin_inline[3];
  const float2* pts;         // packed endpoints
  const int* offsets;        // [B+1] or nullptr (shared scan)
  int shared_n;
  float* out_pose;           // [B*3]
  float* out_cov;            // [B*9] or nullptr
  float* trace;              // nullptr, or [steps*12] per-GN-step record of scan 0 (draw/debug hooks):
                             // {map-frame estimate after the step [3], H of that step [9] col-major}
  unsigned* done_flag;       // nullptr, or a host-visible word that receives done_seq (system-scope release)
  unsigned done_seq;         // after the single-scan results are written: the host polls it instead of
                             // waiting for the end-of-kernel signal
  unsigned* err_flag;        // nullptr, or a host-visible word that receives done_seq when the cooperative matcher's
                             // exchange gave up waiting for a workgroup's record (the pose is then not to be used)
  int xcd_chunk;             // workgroup -> scan mapping (xcd_block): 0 = one contiguous eighth of the batch per XCD,
                             // c > 0 = chunks of c workgroups dealt to the XCDs in turn
  int wg_sync;               // texel-cache form: the waves of a workgroup meet at a barrier before every beam (L1 sharing)
  int coop_mute_block;       // test hook (hsm_debug_set_coop_mute): 1 + the workgroup of the multi-workgroup matcher that never
                             // publishes its records, 0 = none -- how the suite provokes the exchange timeout
  unsigned long long* clock_probe;  // nullptr, or four words the wave of scan 0 fills: shader-clock counter (s_memtime)
                                    // and 100 MHz wall clock at its first GN step [0,1] and at its end [2,3]
  float* spec_scratch;     // gn_match_spec_kernel: [batch][spec_stride] float4s for the nine products of every beam
  unsigned spec_stride;    // float4s per scan
  SpecStats* spec_stats;   // nullptr, or counters the stitching pass adds to (hsm_debug_spec_stats)
  int n_bound;             // HOST ONLY: 0 = scan lengths live on the device only (max_n is a hint), else no scan is longer than this
  const int* perm;         // nullptr, or [batch]: launch slot -> scan index (hsm_set_batch_order: the batch in Morton order of its start
                           // poses; results still land at the scan's own index).  Read by the texel-cache batch forms only
  ExchangeFused xp;        // world > 0: this launch posts its poses into an exchange and unpacks an earlier epoch (forms that support
                           // it say so by setting hsm_ctx::fused_exchange_done; the others leave both to a launch of pose_exchange.hip)
};

__device__ __forceinline__ void publish_done(const MatchParams& P) {
  if (P.done_flag) __hip_atomic_store(P.done_flag, P.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- a matcher launch that takes part in the pose exchange itself (MatchParams::xp, pose_exchange.h) -------------------------------
// POST: the wavefront that has just finished scan `row` stores its pose -- three 8-byte {value, epoch tag} granules per rank, one
// system-scope store each (global_store_dwordx2 sc0 sc1), lanes 0 .. 3 world - 1 in ONE instruction -- into every rank's mailbox.
__device__ __forceinline__ void exchange_post_pose(const ExchangeFused& X, int row, float x, float y, float th) {
  const int l = (int)(threadIdx.x & 63u);
  if (l < 3 * X.world) {
    const int p = l / 3, c = l - 3 * p;
    const float v = c == 0 ? x : (c == 1 ? y : th);
    uint64_t* dst = X.peer[p] + X.post_off + (size_t)row * 3 + c;
    __hip_atomic_store(dst, ((uint64_t)X.post_tag << 32) | (uint64_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// WAIT + unpack: workgroup `wb` of the `X.wait_blocks` extra workgroups behind the matcher's own (they are dispatched as those
// retire, i.e. into the launch's tail, when the epoch they wait for -- one match back -- has long arrived); bounded like the
// stand-alone kernel's wait (pose_exchange.hip)
__device__ __forceinline__ void exchange_wait_unpack(const ExchangeFused& X, int wb) {
  const uint64_t* box = X.peer[X.rank] + X.wait_off;
  unsigned long long t0 = 0;
  for (int i = wb * (int)blockDim.x + (int)threadIdx.x; i < X.total_granules; i += X.wait_blocks * (int)blockDim.x) {
    uint64_t g = __hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    bool late = false;
    while ((uint32_t)(g >> 32) != X.wait_tag) {
      if (t0 == 0) t0 = wall_clock64() | 1ull;
      __builtin_amdgcn_s_sleep(4);
      g = __hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((uint32_t)(g >> 32) != X.wait_tag && wall_clock64() - t0 > X.timeout_ticks) {
        late = true;
        break;
      }
    }
    if (late) {
      __hip_atomic_fetch_add(X.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(X.status + 1, X.wait_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (X.out) X.out[i] = __uint_as_float(late ? 0x7fc00000u : (uint32_t)g);
  }
}

// Workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md; used for speed only -- any placement is
// correct), and every XCD has its own 4 MiB L2.  A batch is usually spatially ordered (consecutive scans of a
// trajectory, hypotheses around one pose).  Two mappings, chosen per launch by the host (MatchParams::xcd_chunk):
//   contiguous (xcd_chunk == 0): each XCD takes one CONTIGUOUS eighth of the batch, so the map region its L2 has to hold
//     is eight times smaller than with the hardware's round robin -- the mapping for maps whose touched region
//     outgrows the L2s (4096^2 pyramid: 135 vs 138 us, exact form 308 vs 319 us);
//   chunked cyclic (xcd_chunk == c): XCD x takes chunks x, x + 8, x + 16, ... of c consecutive workgroups.  How long a
//     scan takes depends on where it was taken (per-wave time stamps: 37.6 .. 44.1 us between sixteenths of the bench
//     batch), so contiguous eighths leave whole XCDs with the slow stretches; chunks keep an XCD's working set
//     compact (16 workgroups = 64 consecutive scans) and give every XCD a sample of the whole batch (2048^2 headline:
//     49.2 vs 50.5 us).
// Bijective for any grid.  -DHSM_XCD_SWIZZLE=0 keeps the hardware order.
#ifndef HSM_XCD_SWIZZLE
#define HSM_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ int xcd_block(int b, int nblocks, int chunk) {
#if HSM_XCD_SWIZZLE
  int base = 0;
  if (chunk > 0) {
    const int main_blocks = nblocks / (8 * chunk) * (8 * chunk);
    if (b < main_blocks) {
      const int xcd = b & 7, j = b >> 3;
      return ((j / chunk) * 8 + xcd) * chunk + j % chunk;
    }
    base = main_blocks;  // the remainder of the grid: contiguous
  }
  const int rb = b - base, rn = nblocks - base;
  const int xcd = rb & 7, idx = rb >> 3, q = rn >> 3, r = rn & 7;
  return base + xcd * q + (xcd < r ? xcd : r) + idx;
#else
  (void)nblocks;
  (void)chunk;
  return b;
#endif
}

// The lane index, recomputed where it is used (two v_mbcnt) instead of being carried in a VGPR from the top of a
// kernel that has none to spare: the volatile asm is neither hoisted nor merged with the threadIdx-derived value.
__device__ __forceinline__ int lane_id_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// Texel address of cell (x, y) in the quad plane.
// HSM_QUAD_TILE == 0 (default): row major, index = y*sizeX + x like the reference's grid -- one
//   v_mad_u32_u24 per beam.
// HSM_QUAD_TILE == 1: 4x2-cell tiles (= eight 16-byte texels = one 128-byte line), Morton order
//   inside (x0, y0, x1), which cuts the cache-line requests of a gather along a wall by about a
//   third but costs five more integer VALU ops per beam.  Measured on MI355X the matcher is
//   VALU-issue bound and both variants run at the same speed (profiles/r01/kernel_ab_tile.jsonl),
//   so the simpler one is the default; the switch is kept for maps that outgrow the L2.
#ifndef HSM_QUAD_TILE
#define HSM_QUAD_TILE 0
#endif
__host__ __device__ __forceinline__ unsigned quad_index(unsigned x, unsigned y, int tiles_x, int sx) {
#if HSM_QUAD_TILE
  (void)sx;
  return ((((y >> 1) * (unsigned)tiles_x) + (x >> 2)) << 3) | ((x & 2) << 1) | ((y & 1) << 1) | (x & 1);
#else
  (void)tiles_x;
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(y, (unsigned)sx) + x;
#else
  return y * (unsigned)sx + x;
#endif
#endif
}

// Transform<Affine> * Vector2f = t + (l(i,0)*x + l(i,1)*y)   (Eigen Transform.h)
__device__ __forceinline__ void affine_apply(const Affine2& a, float x, float y, float& ox, float& oy) {
  ox = a.t0 + (a.l00 * x + a.l01 * y);
  oy = a.t1 + (a.l10 * x + a.l11 * y);
}

// sinf/cosf of the reference (float overloads; OccGridMapUtil.h:70-71 compile to one
// sincosf call): glibc's algorithm operation for operation (libm_exact.h) -- identical bits for every
// argument, ~20 fp64 operations, no table on the |theta| < 120 path the matcher lives on.
// PIN: the texel-cache forms pin the binary64 constants to their use (libm_exact.h at_use: they have no register to hoist
// them into); the latency forms let the optimiser hoist them out of the 14-step chain
template <bool PIN = false>
__device__ __forceinline__ void sincos_f32(float th, float& s, float& c) {
#if defined(HSM_EXPERIMENTS) && defined(HSM_EXP_FAST_SINCOS)
  // what-if build (NOT parity-safe): the hardware's v_sin_f32 / v_cos_f32 -- two instructions -- instead of glibc's binary64
  // evaluation.  An upper bound on what any restructuring of the bit-exact sincosf can buy per Gauss-Newton step
  // (round-3 verdict item 7; profiles/r04/README.md)
  s = __sinf(th);
  c = __cosf(th);
#else
  libm::sincosf_glibc<PIN>(th, s, c);
#endif
}

// util::normalize_angle (HSL/util/UtilFunctions.h:37-49): double fmod, float result
template <bool PIN = false>
__device__ __forceinline__ float normalize_angle(float angle) {
  const double two_pi = libm::at_use<PIN>(2.0 * 3.14159265358979323846);  // PIN: materialised here, not hoisted over the GN loops
  float a = (float)fmod(fmod((double)angle, two_pi) + two_pi, two_pi);
  if ((double)a > libm::at_use<PIN>(3.14159265358979323846)) {
    a = (float)((double)a - two_pi);
  }
  return a;
}

// The LevelView fields the beam loop reads, pulled into registers ONCE per level (the kernel
// argument block lives in memory; re-reading it per beam costs a scalar load + wait each time).
struct LevelRegs {
  const float4* quad;
  const float* prob;
  int sx;
  int tiles_x;
  float limx, limy;
  int zero_index;  // index of the all-zero texel / pad cells behind the plane (see sample_fetch)
};

template <int LAYOUT>
__device__ __forceinline__ LevelRegs level_regs(const LevelView& L) {
  LevelRegs R;
  R.quad = L.quad;
  R.prob = L.prob;
  R.sx = L.sx;
  R.tiles_x = L.tiles_x;
  R.limx = L.limx;
  R.limy = L.limy;
  R.zero_index = LAYOUT == kLayoutQuad ? L.quad_texels : L.sx * L.sy;
  return R;
}

// a1, split in two so that a lane can ISSUE the texel gathers of several beams back to back
// (stage 1) before it CONSUMES any of them (stage 2): memory latency is hidden by
// instruction-level parallelism inside the wavefront, not only by occupancy.
// Two fp32 values kept together (storage only).  Measured on MI355X (tools/microbench/valu_rate.hip,
// profiles/r01/valu_rate.txt): v_mul/v_add/v_fma_f32 issue every ~2.5 cycles per wave64, while
// v_pk_mul/v_pk_add_f32 -- and v_med3, v_fract, v_cvt, v_mad_u32_u24 -- take ~4: a packed op is
// worth 1.6 scalar ones, which the v_mov shuffles needed to form pairs eat up again.  So the
// arithmetic below is plain scalar fp32 and the library is built with -fno-slp-vectorize.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));  // a native 128-bit register tuple (inline-asm operand)

struct BeamSample {
  f2 lo, hi;  // (P(ix,iy), P(ix+1,iy)), (P(ix,iy+1), P(ix+1,iy+1))
  f2 X, Y;    // (xFacInv, fx), (yFacInv, fy)  -- OccGridMapUtil.h:298,338-339
};

// stage 1: bounds test, cell index, fractions, and the (asynchronous) gather.
//
// Out-of-map beams (MapDimensionProperties::pointOutOfMapBounds, MapDimensionProperties.h:65-68)
// must contribute EXACT zeros, like the (0,0,0) the reference returns at OccGridMapUtil.h:290-292.
// They are pointed at an all-zero texel stored behind the plane: P00=P10=P01=P11=0 gives
// dM/dx = dM/dy = -0 and M = 0, so every product added to H and dTr is +-0 and the running fp32
// sums are unchanged bit for bit -- the same outcome as the reference's explicit zeros, without a
// branch or three selects per beam.  The matcher is VALU-issue bound (profiles/r01), so the test
// itself is written for instruction count: v_med3_f32 clamps the coordinate into [0, dims-2]; the
// beam is outside exactly when the clamp changed it (x < 0 or x > dims-2, as in the reference),
// and the clamped value doubles as the finite stand-in coordinate of an outside beam.  A NaN
// coordinate never equals its clamp, so it counts as outside (the reference would index the map
// with (int)NaN there and crash).  v_fract_f32(x) == x - (float)(int)x exactly for 0 <= x < 2^23.
// Bounds test of one map coordinate pair.  HSM_BOUNDS_BITS=1 (default): 0 <= x <= lim on the BIT PATTERNS -- for
// x >= +0 the unsigned pattern of an fp32 value is monotonic in the value, every negative value (sign bit set) and
// every NaN compares above any non-negative limit, so `bits(x) <= bits(lim)` is the whole test in ONE v_cmp_le_u32
// per axis (the clamp form costs v_med3 + v_cmp_neq per axis).  The one value the pattern test would judge
// differently is -0.0 (inside for the reference: -0.0 < 0 is false); c = e + r is -0.0 only if e AND r are -0.0,
// and the callers pass e + 0.0f (wave-uniform, once per GN step; changes no other sum), so it cannot occur.
// Outside beams keep their raw coordinate: v_cvt_i32_f32 saturates and v_fract_f32 of a finite value is finite,
// which is all the all-zero texel needs to yield +-0 products.
#ifndef HSM_BOUNDS_BITS
#define HSM_BOUNDS_BITS 1
#endif
// the estimate's map coordinates as the beam loop adds them to the rotated endpoints: -0.0 -> +0.0 (see above; for
// every other value x + 0.0f == x, and (+0.0) + r == (-0.0) + r unless r is -0.0 too)
__device__ __forceinline__ f2 step_origin(float ex, float ey) {
#if HSM_BOUNDS_BITS
  return f2{ex + 0.0f, ey + 0.0f};
#else
  return f2{ex, ey};
#endif
}

struct CellCoord {
  unsigned ix, iy;
  float fx, fy;
  bool oob;
};

__device__ __forceinline__ CellCoord cell_coord(const LevelRegs& L, f2 c) {
  CellCoord q;
#if HSM_BOUNDS_BITS
  q.oob = (int)(__float_as_uint(c.x) > __float_as_uint(L.limx)) | (int)(__float_as_uint(c.y) > __float_as_uint(L.limy));
  const float sx_ = c.x, sy_ = c.y;
#else
  const float sx_ = __builtin_amdgcn_fmed3f(c.x, 0.0f, L.limx);
  const float sy_ = __builtin_amdgcn_fmed3f(c.y, 0.0f, L.limy);
  q.oob = (sx_ != c.x) | (sy_ != c.y);
#endif
#if HSM_BOUNDS_BITS
  // truncation, OccGridMapUtil.h:295.  The instruction itself (saturating, defined for every input) rather than a
  // C++ cast, whose result is undefined for the raw coordinate of an outside beam.
  asm("v_cvt_i32_f32 %0, %1" : "=v"(q.ix) : "v"(sx_));
  asm("v_cvt_i32_f32 %0, %1" : "=v"(q.iy) : "v"(sy_));
#else
  q.ix = (unsigned)(int)sx_;  // truncation, OccGridMapUtil.h:295
  q.iy = (unsigned)(int)sy_;
#endif
  q.fx = __builtin_amdgcn_fractf(sx_);  // :298
  q.fy = __builtin_amdgcn_fractf(sy_);
  return q;
}

template <int LAYOUT>
__device__ __forceinline__ BeamSample sample_fetch(const LevelRegs& L, f2 c) {
  BeamSample b;
  const CellCoord q = cell_coord(L, c);
  const bool oob = q.oob;
  const unsigned ix = q.ix, iy = q.iy;
  b.X.y = q.fx;
  b.Y.y = q.fy;
  b.X.x = 1.0f - b.X.y;  // :338-339
  b.Y.x = 1.0f - b.Y.y;
  if (LAYOUT == kLayoutQuad) {
    const unsigned index = oob ? (unsigned)L.zero_index : quad_index(ix, iy, L.tiles_x, L.sx);
    // 32-bit byte offset on a uniform base: one global_load_dwordx4 with an SGPR base address
    const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(L.quad) + (size_t)(index << 4));
    b.lo = f2{q.x, q.y};
    b.hi = f2{q.z, q.w};
  } else {
    // indices index, index+1, index+sizeX, index+sizeX+1 (:302-330); the plane is followed by
    // sizeX+2 zero cells so that the zero_index footprint is all zeros too
    const unsigned index = oob ? (unsigned)L.zero_index : __umul24(iy, (unsigned)L.sx) + ix;
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(L.prob) + (size_t)(index << 2));
    b.lo = f2{p[0], p[1]};
    b.hi = f2{p[L.sx], p[L.sx + 1]};
  }
  return b;
}

// stage 2: the interpolation and the source-literal "derivatives" (:332-346):
//   M  = ((P00*xFacInv + P10*fx) * yFacInv) + ((P01*xFacInv + P11*fx) * fy)
//   gx = -((P00-P10)*xFacInv + (P01-P11)*fx)      gy = -((P00-P01)*yFacInv + (P10-P11)*fy)
// (x-differences blended with the x fractions, as the source does -- SURVEY.md row a1).
// Returned as M and G = (-gx, -gy): the two negations of the source are exact sign flips, and
// every consumer below absorbs them into an operand-negate modifier or a product of two of them.
struct BeamTerms {
  float M;
  f2 G;  // (-dM/dx, -dM/dy) as the source defines them
};

__device__ __forceinline__ BeamTerms sample_finish(const BeamSample& b) {
  const float xFacInv = b.X.x, fx = b.X.y, yFacInv = b.Y.x, fy = b.Y.y;
  const float i0 = b.lo.x, i1 = b.lo.y, i2 = b.hi.x, i3 = b.hi.y;
  const float dx1 = i0 - i1;  // :332-336
  const float dx2 = i2 - i3;
  const float dy1 = i0 - i2;
  const float dy2 = i1 - i3;
  BeamTerms r;
  r.M = ((i0 * xFacInv + i1 * fx) * (yFacInv)) + ((i2 * xFacInv + i3 * fx) * (fy));  // :341-344
  r.G.x = ((dx1 * xFacInv) + (dx2 * fx));  // -gx  :345
  r.G.y = ((dy1 * yFacInv) + (dy2 * fy));  // -gy  :346
  return r;
}

// per-beam contribution to (dTr, H) -- OccGridMapUtil.h:76-98; paired accumulators
struct Acc9 {
  f2 d01;   // dTr[0], dTr[1]
  f2 hd;    // H(0,0), H(1,1)
  f2 hr;    // H(0,2), H(1,2)
  float d2, h22, h01;
  __device__ __forceinline__ void zero() {
    d01 = hd = hr = f2{0.0f, 0.0f};
    d2 = h22 = h01 = 0.0f;
  }
};

// The rotated endpoint R(theta) * p, shared by the transform and by rotDeriv:
//   transform * currPoint   = t + [c -s; s c] p = (ex + (c*px + (-s)*py), ey + (s*px + c*py))   (:80)
//   rotDeriv (:87)          = (-s*px - c*py) * gx + (c*px - s*py) * gy
// In IEEE arithmetic (-s)*py == -(s*py) and a + (-b) == a - b exactly, and round-to-nearest is
// symmetric, so  c*px + (-s)*py == c*px - s*py == rx  and  -s*px - c*py == -(s*px + c*py) == -ry
// BIT FOR BIT: the two products/one sum per component are computed once and reused.
struct BeamRot {
  f2 r;  // (rx, ry)
};

// cs = (cos, sin), sc = (sin, cos) of the current estimate; e = (ex, ey); p = endpoint
template <int LAYOUT>
__device__ __forceinline__ BeamSample beam_fetch(const LevelRegs& L, f2 e, f2 cs, f2 sc, f2 p, BeamRot& r) {
  r.r.x = cs.x * p.x - sc.x * p.y;  // c*px - s*py
  r.r.y = cs.y * p.x + sc.y * p.y;  // s*px + c*py
  return sample_fetch<LAYOUT>(L, f2{e.x + r.r.x, e.y + r.r.y});
}

// With g = -G (the source's gx, gy) and rd = rotDeriv, in IEEE arithmetic:
//   rd      = (-ry)*gx + rx*gy = ry*Gx - rx*Gy          (sign flips are exact)
//   dTr    += g*funVal, rd*funVal  ==  dTr -= G*funVal  (a + (-b) == a - b)
//   H      += g*g, gx*gy           ==  G*G, Gx*Gy
//   H(.,2) += g*rd                 ==  H(.,2) -= G*rd
__device__ __forceinline__ float beam_finish(const BeamSample& b, const BeamRot& r, Acc9& a,
                                             BeamTerms* terms_out = nullptr) {
  const BeamTerms t = sample_finish(b);
  const float Gx = t.G.x, Gy = t.G.y;
  const float funVal = 1.0f - t.M;
  const float rotDeriv = r.r.y * Gx - r.r.x * Gy;  // :87
  a.d01.x -= Gx * funVal;
  a.d01.y -= Gy * funVal;
  a.d2 += rotDeriv * funVal;
  a.hd.x += Gx * Gx;
  a.hd.y += Gy * Gy;
  a.h22 += rotDeriv * rotDeriv;
  a.h01 += Gx * Gy;
  a.hr.x -= Gx * rotDeriv;
  a.hr.y -= Gy * rotDeriv;
  if (terms_out) *terms_out = t;
  return rotDeriv;
}

template <int LAYOUT>
__device__ __forceinline__ float beam_accumulate(const LevelRegs& L, float ex, float ey, float sinRot,
                                                 float cosRot, float px, float py, Acc9& a,
                                                 BeamTerms* terms_out = nullptr) {
  BeamRot r;
  const BeamSample b = beam_fetch<LAYOUT>(L, step_origin(ex, ey), f2{cosRot, sinRot}, f2{sinRot, cosRot}, f2{px, py}, r);
  return beam_finish(b, r, a, terms_out);
}

// ---- HSM_PARITY_EXACT: the reference's summation ORDER -------------------------------------------
// getCompleteHessianDerivs (OccGridMapUtil.h:76-98) runs nine independent fp32 chains over the beams,
// acc_t = acc_t + product_t(i) for i = 0 .. n-1.  The products are the same bits in every form of the
// matcher; what the throughput forms change is the order of the additions (lane-strided partial sums +
// tree), which is where the last-bit differences of H and dTr -- and, on scans where Gauss-Newton has
// not settled, the visible pose differences -- come from.  The exact form keeps the reference's order:
// a round of T consecutive beams (one per lane of the team) writes its 9 x T products to LDS, row t =
// term t, and lane t of the team's first wave (t = 0..8) adds row t to its running sum left to right.
// The nine chains run side by side in nine lanes; a round costs 64 dependent v_add_f32 per 64 beams on
// top of the beam arithmetic (about 2.5x the instruction count of the fast form).  Padding lanes and
// out-of-map beams contribute +-0, which leaves a running sum that started at +0 unchanged bit for bit.
constexpr int kExactGroupRounds = 5;  // rounds the exact-order team form fetches, stages and sums together; a scan of at most that
                                      // many rounds keeps its endpoints in registers (xq_resident): it is read from memory ONCE
constexpr int kExactPad = 4;  // row stride T + 4 floats: rows stay 16-byte aligned, the nine chain lanes hit distinct banks

// the nine products with the reference's signs (g = -G): dTr[0..2], H(0,0), H(1,1), H(2,2), H(0,1), H(0,2), H(1,2)
__device__ __forceinline__ void beam_products(const BeamSample& b, const BeamRot& r, float pr[9]) {
  const BeamTerms t = sample_finish(b);
  const float Gx = t.G.x, Gy = t.G.y;
  const float funVal = 1.0f - t.M;
  const float rotDeriv = r.r.y * Gx - r.r.x * Gy;  // :87
  pr[0] = -(Gx * funVal);
  pr[1] = -(Gy * funVal);
  pr[2] = rotDeriv * funVal;
  pr[3] = Gx * Gx;
  pr[4] = Gy * Gy;
  pr[5] = rotDeriv * rotDeriv;
  pr[6] = Gx * Gy;
  pr[7] = -(Gx * rotDeriv);
  pr[8] = -(Gy * rotDeriv);
}

// one round: beams base .. base+T-1 (thread tid holds beam base+tid).  `run` is meaningful in threads
// 0..8 of the team only.  T == 64: one wavefront, whose LDS operations execute in program order -- no
// barrier; T > 64: the team owns its workgroup and synchronises around the chain.
// `count` = beams of this round that exist (the lanes beyond stage +-0 products, which leave a running sum unchanged bit for
// bit, so the chain stops after the last float4 pair that holds a real beam).  The row streams through two 32-byte halves:
// a half is requested again right after its values are consumed, so a dependent addition costs its own latency (8.5 cycles
// on a lone wavefront, tools/ubench_chain.hip) instead of 14.4 with one 16-byte read in flight -- the chain is 1081 x 14
// additions of a single-scan match in exact mode, ~100 of its ~135 us before.
// stage the nine products of the beam at position `pos` of rows of RL beams
template <int RL>
__device__ __forceinline__ void exact_stage(const float pr[9], float* __restrict__ stage, int pos) {
#pragma unroll
  for (int t = 0; t < 9; ++t) stage[t * (RL + kExactPad) + pos] = pr[t];
}

// the chains over the first `count` beams of the staged rows (rows of RL beams; SYNC: the team is wider than one wavefront)
template <int RL, bool SYNC>
__device__ __forceinline__ float exact_chain(float* __restrict__ stage, int tid, float run, int count) {
  if (SYNC) __syncthreads();
  if (tid < 9) {
    const f4v* row = reinterpret_cast<const f4v*>(stage + tid * (RL + kExactPad));
    // 16 beams per iteration, branch-free (loads behind a branch make the compiler wait for all of them): the last
    // iteration's read-ahead is clamped into the row, beams between `count` and the next multiple of 16 add their +-0
    const int nq = ((count + 15) >> 4) << 2;  // float4s, a multiple of 4, <= RL / 4
    f4v a0 = row[0], a1 = row[1], b0 = row[2], b1 = row[3];  // two 32-byte halves in flight
    // (the additions as inline asm: left to itself the compiler keeps the running sum in the registers of the half it is
    // consuming, which postpones that half's refill to the end of the iteration -- one exposed LDS round trip per 16 beams)
    auto add8 = [&](const f4v& u, const f4v& v) {
      asm volatile(
          "v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0\n\t"
          "v_add_f32 %0, %5, %0\n\tv_add_f32 %0, %6, %0\n\tv_add_f32 %0, %7, %0\n\tv_add_f32 %0, %8, %0"
          : "+v"(run)
          : "v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)
          : "memory");
    };
    for (int q = 0; q < nq; q += 4) {
      add8(a0, a1);
      const int qa = min(q + 4, RL / 4 - 4);  // refilled right behind its last use: eight additions to arrive in
      a0 = row[qa], a1 = row[qa + 1];
      add8(b0, b1);
      b0 = row[qa + 2], b1 = row[qa + 3];
    }
  }
  if (SYNC) __syncthreads();
  return run;
}

// one round of T beams: stage, sum
template <int T>
__device__ __forceinline__ float exact_round(const float pr[9], float* __restrict__ stage, int tid, float run, int count = T) {
  exact_stage<T>(pr, stage, tid);
  return exact_chain<T, (T > 64)>(stage, tid, run, count);
}

// Wavefront all-reduce without LDS traffic: four DPP steps inside each row of 16 lanes (the
// cross-lane operand rides on the v_add itself), then the two gfx950 row/half swaps
// (v_permlane16_swap, v_permlane32_swap).  Every lane ends with the same bits.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float wave_allreduce(float v) {
  v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_f32<0x141>(v);  // row_half_mirror: the other quad of the 8-lane half row
  v += dpp_f32<0x140>(v);  // row_mirror: the other half of the 16-lane row
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = __int_as_float(r[0]) + __int_as_float(r[1]);  // rows 0+1, 2+3
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    v = __int_as_float(r[0]) + __int_as_float(r[1]);  // lower + upper 32 lanes
  }
  return v;
}

// The nine totals at once, "folded": a butterfly level that pairs lanes i and p(i) needs the sum of a value only in
// ONE lane of each pair if the other lane of the pair carries a second value -- so each level halves the number of
// live registers instead of keeping nine.  v_permlane32_swap / v_permlane16_swap do exactly that for two registers
// (swap + add = 2 instructions per PAIR of values instead of mov + swap + add per value), the two row levels do it with
// DPP bank masks (v_add_f32_dpp writes only the enabled banks of its destination), and the last two levels run on
// the single register that is left.  25 instructions + 9 v_readlane_b32 (the totals end up in SGPRs, identical in
// every lane) instead of 90.  Level order 32, 16, row_mirror, row_half_mirror, lane^2, lane^1 -- the same pairing
// tree for all nine values.
#ifndef HSM_REDUCE_FOLD
#define HSM_REDUCE_FOLD 1
#endif

__device__ __forceinline__ float fold_swap32(float a, float b) {  // lanes 0..31: a[i] + a[i+32];  32..63: b[i-32] + b[i]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float fold_swap16(float a, float b) {  // rows 0, 2: a (row + row^1);  rows 1, 3: b
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}

__device__ __forceinline__ void wave_allreduce9_folded(Acc9& a) {
  // level 32 and level 16: the totals over the four rows, per column of 16
  const float a0 = fold_swap32(a.d01.x, a.d01.y);  // rows 0,1: dTr0   rows 2,3: dTr1
  const float a1 = fold_swap32(a.d2, a.hd.x);      //           dTr2             H00
  const float a2 = fold_swap32(a.hd.y, a.h22);     //           H11              H22
  const float a3 = fold_swap32(a.h01, a.hr.x);     //           H01              H02
  const float a4 = fold_swap32(a.hr.y, a.hr.y);    // H12 everywhere
  float b0 = fold_swap16(a0, a1);                  // rows: dTr0, dTr2, dTr1, H00
  float b1 = fold_swap16(a2, a3);                  // rows: H11,  H01,  H22,  H02
  float b2 = fold_swap16(a4, a4);                  // H12 in every row
  // row levels.  s_nop: the hazard recogniser does not look into inline asm (a DPP operand needs two wait states
  // after the VALU write of its register)
  asm("s_nop 1\n\t"
      // row_mirror (lane i <-> 15 - i): banks 2,3 of b1 keep b1's sums, banks 0,1 of b1 take b0's; b2 plain
      "v_add_f32_dpp %[b1], %[b1], %[b1] row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %[b1], %[b0], %[b0] row_mirror row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %[b2], %[b2], %[b2] row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      // row_half_mirror (i <-> 7 - i inside each half row): banks 1,3 of b2 keep H12, banks 0,2 take b1's
      "v_add_f32_dpp %[b2], %[b2], %[b2] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %[b2], %[b1], %[b1] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %[b2], %[b2], %[b2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %[b2], %[b2], %[b2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : [b1] "+v"(b1), [b2] "+v"(b2)
      : [b0] "v"(b0));
  // where the totals sit: lane 16 * row + 4 * bank.  bank 0 <- b0's rows, bank 2 <- b1's rows, banks 1, 3 <- H12
  auto lane_value = [&](int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b2), lane)); };
  a.d01.x = lane_value(0);
  a.d2 = lane_value(16);
  a.d01.y = lane_value(32);
  a.hd.x = lane_value(48);
  a.hd.y = lane_value(8);
  a.h01 = lane_value(24);
  a.h22 = lane_value(40);
  a.hr.x = lane_value(56);
  a.hr.y = lane_value(4);
}

__device__ __forceinline__ void wave_allreduce9(Acc9& a) {
#if HSM_REDUCE_FOLD
  wave_allreduce9_folded(a);
  return;
#endif
  a.d01.x = wave_allreduce(a.d01.x);
  a.d01.y = wave_allreduce(a.d01.y);
  a.d2 = wave_allreduce(a.d2);
  a.hd.x = wave_allreduce(a.hd.x);
  a.hd.y = wave_allreduce(a.hd.y);
  a.h22 = wave_allreduce(a.h22);
  a.h01 = wave_allreduce(a.h01);
  a.hr.x = wave_allreduce(a.hr.x);
  a.hr.y = wave_allreduce(a.hr.y);
}

// team-wide totals: wave all-reduce, then (WPS > 1) LDS staging of the per-wave partials; every thread of the team
// returns with identical bits.  The stage is TRANSPOSED, red[buf][term][wave]: after the barrier lane t (t < 9) reads the
// WPS partials of term t as 16-byte rows and adds them in wave order (the same order as ever: w = 0, 1, 2, ...), and nine
// v_readlane hand the totals to every lane as SGPRs -- 1 + 3 + 9 instructions for four waves instead of the 36 LDS reads
// and 27 additions every lane used to do on its own (the exchange sits on the 14-step dependent chain of a single-scan
// match: its cost grows with WPS, which is what held wider teams back).
template <int WPS>
__device__ __forceinline__ void team_allreduce9(Acc9& a, float (*red)[9][WPS < 4 ? 4 : WPS], int buf, int wave_in_team,
                                                int lane) {
  wave_allreduce9(a);
  if (WPS > 1) {
    if (lane == 0) {
      float(*r)[WPS < 4 ? 4 : WPS] = red[buf];
      r[0][wave_in_team] = a.d01.x; r[1][wave_in_team] = a.d01.y; r[2][wave_in_team] = a.d2;
      r[3][wave_in_team] = a.hd.x; r[4][wave_in_team] = a.hd.y; r[5][wave_in_team] = a.h22;
      r[6][wave_in_team] = a.h01; r[7][wave_in_team] = a.hr.x; r[8][wave_in_team] = a.hr.y;
    }
    __syncthreads();
    const float* row = red[buf][lane < 9 ? lane : 8];
    float t = row[0];
#pragma unroll
    for (int w = 1; w < WPS; ++w) t += row[w];
    auto total = [&](int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), k)); };
    a.d01 = f2{total(0), total(1)}; a.d2 = total(2);
    a.hd = f2{total(3), total(4)}; a.h22 = total(5);
    a.h01 = total(6); a.hr = f2{total(7), total(8)};
  }
}

// H.inverse() * dTr as Eigen evaluates it (LU/InverseImpl.h cofactors * invdet, then a
// coefficient-based product; 3-term sums are x0 + (x1 + x2)), ScanMatcher.h:201-217.
__device__ __forceinline__ void gn_solve_and_step(const Acc9& a, float& ex, float& ey, float& eth) {
  if ((a.hd.x != 0.0f) && (a.hd.y != 0.0f)) {
    // symmetric H: m(r,c)
    const float m00 = a.hd.x, m01 = a.h01, m02 = a.hr.x;
    const float m10 = a.h01, m11 = a.hd.y, m12 = a.hr.y;
    const float m20 = a.hr.x, m21 = a.hr.y, m22 = a.h22;
    // cofactor_3x3<i,j> = m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1), i1=(i+1)%3 ...
    const float c00 = m11 * m22 - m12 * m21;
    const float c10 = m21 * m02 - m22 * m01;
    const float c20 = m01 * m12 - m02 * m11;
    const float det = c00 * m00 + (c10 * m10 + c20 * m20);
    const float invdet = 1.0f / det;
    const float i00 = c00 * invdet, i01 = c10 * invdet, i02 = c20 * invdet;
    const float i10 = (m12 * m20 - m10 * m22) * invdet;  // cofactor<0,1>
    const float i11 = (m22 * m00 - m20 * m02) * invdet;  // cofactor<1,1>
    const float i12 = (m02 * m10 - m00 * m12) * invdet;  // cofactor<2,1>
    const float i20 = (m10 * m21 - m11 * m20) * invdet;  // cofactor<0,2>
    const float i21 = (m20 * m01 - m21 * m00) * invdet;  // cofactor<1,2>
    const float i22 = (m00 * m11 - m01 * m10) * invdet;  // cofactor<2,2>
    const float s0 = i00 * a.d01.x + (i01 * a.d01.y + i02 * a.d2);
    const float s1 = i10 * a.d01.x + (i11 * a.d01.y + i12 * a.d2);
    float s2 = i20 * a.d01.x + (i21 * a.d01.y + i22 * a.d2);
    if (s2 > 0.2f) {
      s2 = 0.2f;
    } else if (s2 < -0.2f) {
      s2 = -0.2f;
    }
    ex += s0;
    ey += s1;
    eth += s2;
  }
}

// The SIMD's issue arbiter favours its oldest wavefront.  In a throughput launch -- one wavefront per scan, four per
// SIMD, all started within 2 us -- the four therefore finish one after another (time stamps of a headline launch:
// ~33 / 42 / 50 / 52 us) and the last ones run with too few neighbours to cover their gathers.  Rotating the priority
// from GN step to GN step (slot id of the wave + step counter, two SALU instructions) keeps them in step: headline
// 57.1 -> 55.4 us, 3-level pyramid 118.7 -> 110 us, 4096^2 pyramid 150 -> 142 us (profiles/r02/README.md).  Used by the
// texel-cache form only: the form that gathers every beam in every step loses 4 % with it (67 -> 69.5 us).
__device__ __forceinline__ void rotate_wave_priority(int step) {
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));  // bits 3:0 = wave slot on its SIMD
  switch (((hwid & 0xFu) + (unsigned)step) & 3u) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// One team (WPS wavefronts) per scan; SPB scans per workgroup (SPB > 1 only when WPS == 1,
// where no barrier is ever executed so the wavefronts of a block are fully independent).
//
// BPL > 0: "beams per lane" register-resident form.  A scan with n <= 64*WPS*BPL beams is loaded
// ONCE (coalesced float2, beam i -> slot i / (64*WPS) of lane i mod 64*WPS) and stays in VGPRs for
// all levels and GN steps; the beam loop is fully unrolled so the BPL texel gathers of a step are
// independent and issue back to back (latency hiding by ILP, not only by occupancy).  The
// per-lane summation order (ascending beam index) is the same as the memory loop's, so both
// forms produce identical bits.  Longer scans (or BPL == 0) take the memory loop.
// Occupancy the kernel is BUILT for (and test_kernel_resources.py asserts): four waves per SIMD, except the exact-order teams
// of two and four wavefronts -- they keep a whole group of rounds alive at once (five endpoints, five texels, five rotated
// points per lane: ~150 VGPRs) and stage it in 23 / 46 KB of LDS, and they are latency launches (one team per scan, chosen only
// when the batch cannot fill the chip), so three waves per SIMD is what they get and what they ask for.
template <int WPS, int SPB, int LAYOUT, int BPL, bool EXACT = false>
__global__ void __launch_bounds__(64 * WPS * SPB, ((EXACT && SPB == 1 && (WPS == 2 || WPS == 4)) ? 3 : 4)) gn_match_kernel(const MatchParams P) {
  static_assert(WPS == 1 || SPB == 1, "barrier-synchronised teams own their workgroup");
  static_assert(!EXACT || BPL == 0, "the exact-order form streams the endpoints");
  constexpr int T = 64 * WPS;  // lanes per team
  __shared__ __attribute__((aligned(16))) float red[2][9][WPS < 4 ? 4 : WPS];
  // exact order: [team][term][beam]; single-scan teams of up to four wavefronts stage a whole GROUP of rounds (kXGroup x T beams,
  // 46 KB at four wavefronts) and sum it in one go, the others round by round
#ifndef HSM_XTEAM8  // experiment: an eight-wavefront team stages its THREE rounds (1536 beams, 55 KB) together as well
#define HSM_XTEAM8 0
#endif
  constexpr bool kTeam8 = HSM_XTEAM8 != 0 && EXACT && SPB == 1 && T == 512;
  constexpr int kXGroup = kTeam8 ? 3 : kExactGroupRounds;
  constexpr int kXStaged = (EXACT && SPB == 1 && (T <= 256 || kTeam8)) ? kXGroup : 1;  // rounds staged together
  constexpr int kXRowLen = kXStaged * T;
  __shared__ float stage[EXACT ? SPB * 9 * (kXRowLen + kExactPad) : 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int team = wave / WPS;
  const int wit = wave - team * WPS;
  const int scan = __builtin_amdgcn_readfirstlane(xcd_block((int)blockIdx.x, (int)gridDim.x, P.xcd_chunk) * SPB + team);  // wave-uniform
  if (scan >= P.batch) return;  // whole team exits together
#ifdef HSM_TEAM_TIMELINE
  if (EXACT && P.clock_probe != nullptr && scan == 0 && threadIdx.x == 0) P.clock_probe[4] = wall_clock64();
#endif

  int beg = 0, n = P.shared_n;
  if (P.offsets) {
    beg = P.offsets[scan];
    n = P.offsets[scan + 1] - beg;
  }
  float pw0, pw1, pw2;
  if (P.begin_world) {
    pw0 = P.begin_world[3 * scan + 0];
    pw1 = P.begin_world[3 * scan + 1];
    pw2 = P.begin_world[3 * scan + 2];
  } else {  // single-scan host entry: the pose rides in the kernel arguments, no H2D copy
    pw0 = P.begin_inline[0];
    pw1 = P.begin_inline[1];
    pw2 = P.begin_inline[2];
  }
  if (n == 0) {  // ScanMatcher.h:68,189: pose passes through, cov untouched
    if (lane == 0 && wit == 0) {
      P.out_pose[3 * scan + 0] = pw0;
      P.out_pose[3 * scan + 1] = pw1;
      P.out_pose[3 * scan + 2] = pw2;
      if (scan == 0) publish_done(P);
    }
    return;
  }
  const float2* __restrict__ pts = P.pts + beg;
  const int tid_in_team = wit * 64 + lane;
  constexpr int NREG = BPL > 0 ? BPL : 1;
  f2 pt[NREG];
  const bool in_regs = BPL > 0 && n <= T * BPL;  // team-uniform
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int i = tid_in_team + k * T;
      // padding slots of the last partial row: an endpoint far outside ANY map.  |R(theta) p| =
      // |p| ~ 1.4e30 for every theta, so the transformed point is out of bounds on at least one
      // axis and takes the exact-zero path of sample_fetch; all intermediates stay finite and the
      // per-level power-of-two rescale keeps it huge.
      const float2 q = i < n ? pts[i] : make_float2(1.0e30f, 1.0e30f);
      pt[k] = f2{q.x, q.y};
    }
  }
  // exact order: a scan that is ONE group of rounds keeps its endpoints (unscaled) in registers across all levels and GN steps
  // -- one dependent memory round trip per step, the texel gather, instead of two
  const bool xq_resident = EXACT && n <= kXGroup * T;  // (team-uniform)
  float2 xq[EXACT ? kXGroup : 1];
  if (xq_resident) {
#pragma unroll
    for (int g = 0; g < (EXACT ? kXGroup : 1); ++g) {
      const int i = g * T + tid_in_team;
      xq[g] = i < n ? pts[i] : make_float2(1.0e30f, 1.0e30f);
    }
  }
  Acc9 acc;
  acc.zero();
  int buf = 0;
  int step = 0;
  float reg_scale = 1.0f;  // scale the register-resident endpoints currently carry
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);  // getMapCoordsPose, GridMapBase.h:235-239
    eth = pw2;
    const float ps = L.pt_scale;
    const int gn_steps = L.gn_steps;
    const LevelRegs R = level_regs<LAYOUT>(L);
    if (in_regs) {
      // DataContainer::setFrom(scan, 2^-level): rescale IN PLACE when the level changes.  All
      // factors are powers of two, so p*2^-a*2^(a-b) == p*2^-b bit for bit, and no second
      // register copy of the scan is kept alive across the GN steps.
      const float ratio = ps / reg_scale;
      reg_scale = ps;
#pragma unroll
      for (int k = 0; k < NREG; ++k) pt[k] *= f2{ratio, ratio};
    }
    for (int it = 0; it < gn_steps; ++it) {
#ifdef HSM_TEAM_TIMELINE  // (variant builds, tools/study/team_phase_probe.py: cycles of a GN step's phases, summed over the steps)
      const unsigned long long tt0 = __builtin_readcyclecounter();
      unsigned long long tt1 = tt0, tt2 = tt0;
#endif
      float sinRot, cosRot;
      sincos_f32(eth, sinRot, cosRot);
      acc.zero();
      const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
      if (in_regs) {
#if HSM_PIPELINE
        // Software pipeline of depth one (experiment switch): the texel gather of beam k+1 is issued
        // BEFORE beam k is consumed.  The accumulation order stays k = 0, 1, 2 ... so the bits equal
        // the memory loop's.
        BeamRot rot_cur, rot_nxt;
        BeamSample cur = beam_fetch<LAYOUT>(R, e2, cs, sc, pt[0], rot_cur), nxt;
#pragma unroll
        for (int k = 0; k < NREG; ++k) {
          if (k + 1 < NREG) nxt = beam_fetch<LAYOUT>(R, e2, cs, sc, pt[k + 1], rot_nxt);
          beam_finish(cur, rot_cur, acc);
          // Pin: the accumulators are final here ("+v") and no later gather may be hoisted above
          // this point ("memory") -- exactly one texel in flight while one is being consumed.
          asm volatile(""
                       : "+v"(acc.d01), "+v"(acc.d2), "+v"(acc.hd), "+v"(acc.h22), "+v"(acc.h01), "+v"(acc.hr)
                       :
                       : "memory");
          cur = nxt;
          rot_cur = rot_nxt;
        }
#else
        // chunks of kUnroll beams: issue all gathers of a chunk, then consume them in beam order
        // (the accumulation order stays k = 0, 1, 2 ... so the bits equal the memory loop's)
        // Throughput launches (one wave per scan, 4 waves per SIMD): one beam at a time is fastest.
        // Latency launches (WPS > 1 is only chosen when the batch cannot fill the chip, typically one wave
        // per SIMD): nobody else hides the L2 latency, so the lane's gathers are issued in chunks before the
        // first is consumed -- all of them up to 5 beams per lane (single 1081-beam scan: kernel 31.5 -> 25 us),
        // else 4 at a time (16k-beam scan on 16 waves: 170 / 155 / 173 us for chunks of 1 / 4 / 9).
        constexpr int kChunk = WPS > 1 ? (NREG <= 5 ? NREG : 4) : kUnroll;
#pragma unroll
        for (int k0 = 0; k0 < NREG; k0 += kChunk) {
          BeamSample smp[kChunk];
          BeamRot rot[kChunk];
#pragma unroll
          for (int u = 0; u < kChunk; ++u) {
            if (k0 + u < NREG)
              smp[u] = beam_fetch<LAYOUT>(R, e2, cs, sc, pt[k0 + u], rot[u]);
          }
#pragma unroll
          for (int u = 0; u < kChunk; ++u) {
            if (k0 + u < NREG) beam_finish(smp[u], rot[u], acc);
          }
          // Pin the chunk: the accumulators must be final here ("+v") and no later gather may be
          // hoisted above this point ("memory").  Without it the compiler issues all BPL gathers
          // first and spills their results; with it at most kUnroll texels are in flight per lane.
          asm volatile(""
                       : "+v"(acc.d01), "+v"(acc.d2), "+v"(acc.hd), "+v"(acc.h22), "+v"(acc.h01), "+v"(acc.hr)
                       :
                       : "memory");
        }
#endif
      } else if (EXACT) {
        float run = 0.0f;
        float* st = stage + team * 9 * (kXRowLen + kExactPad);
        // Rounds in groups of kXGroup: the endpoints of the whole group are requested together, then its texels, then the
        // rounds are summed one after the other -- two memory round trips per group instead of two per round (a 1081-beam
        // scan on four wavefronts is ONE group per GN step; the dependent loads were ~40 % of the exact single-scan match).
        for (int base0 = 0; base0 < n; base0 += kXGroup * T) {  // team-uniform trip count
          float2 q[kXGroup];
#pragma unroll
          for (int g = 0; g < kXGroup; ++g) {
            if (xq_resident) {  // a scan of one group: its endpoints were loaded once, before the first level
              q[g] = xq[g];
            } else {
              const int i = base0 + g * T + tid_in_team;
              q[g] = i < n ? pts[i] : make_float2(1.0e30f, 1.0e30f);  // padding: exact +-0 products (see above)
            }
          }
          BeamSample b[kXGroup];
          BeamRot r[kXGroup];
#pragma unroll
          for (int g = 0; g < kXGroup; ++g) b[g] = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{q[g].x * ps, q[g].y * ps}, r[g]);
          if (kXStaged == kXGroup) {
            // the whole group behind ONE barrier pair, the chain runs through it without a stop (rounds beyond the scan
            // stage +-0 and are not summed)
#pragma unroll
            for (int g = 0; g < kXGroup; ++g) {
              float pr[9];
              beam_products(b[g], r[g], pr);
              exact_stage<kXRowLen>(pr, st, g * T + tid_in_team);
            }
#ifdef HSM_TEAM_TIMELINE
            tt1 = __builtin_readcyclecounter();
#endif
            run = exact_chain<kXRowLen, (T > 64)>(st, tid_in_team, run, min(kXRowLen, n - base0));
#ifdef HSM_TEAM_TIMELINE
            tt2 = __builtin_readcyclecounter();
#endif
          } else {
#pragma unroll
            for (int g = 0; g < kXGroup; ++g) {
              const int base = base0 + g * T;
              if (base < n) {  // team-uniform
                float pr[9];
                beam_products(b[g], r[g], pr);
                run = exact_round<T>(pr, st, tid_in_team, run, min(T, n - base));
              }
            }
          }
        }
        float t[9];
        if (WPS == 1) {
#pragma unroll
          for (int k = 0; k < 9; ++k) t[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(run), k));
        } else {
          // threads 0..8 publish; the next write of red[] lies behind the >= 2 barriers of the next step's rounds
          if (tid_in_team < 9) (&red[0][0][0])[tid_in_team] = run;
          __syncthreads();
#pragma unroll
          for (int k = 0; k < 9; ++k) t[k] = (&red[0][0][0])[k];
        }
        acc.d01 = f2{t[0], t[1]}; acc.d2 = t[2];
        acc.hd = f2{t[3], t[4]}; acc.h22 = t[5];
        acc.h01 = t[6]; acc.hr = f2{t[7], t[8]};
      } else {
        for (int i = tid_in_team; i < n; i += T) {
          const float2 p = pts[i];
          BeamRot r;
          const BeamSample b = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p.x * ps, p.y * ps}, r);
          beam_finish(b, r, acc);
        }
      }
      if (!EXACT) {
        team_allreduce9<WPS>(acc, red, buf, wit, lane);
        buf ^= 1;
      }
      gn_solve_and_step(acc, ex, ey, eth);
#ifdef HSM_TEAM_TIMELINE
      if (EXACT && P.clock_probe != nullptr && scan == 0 && tid_in_team == 0) {
        const unsigned long long tt3 = __builtin_readcyclecounter();
        const bool first = l == P.first_level && it == 0;
        if (first) P.clock_probe[8] = P.clock_probe[9] = P.clock_probe[10] = P.clock_probe[11] = 0, P.clock_probe[12] = tt0, P.clock_probe[13] = wall_clock64();
        const int sno = (int)P.clock_probe[11];
        P.clock_probe[8] += tt1 - tt0, P.clock_probe[9] += tt2 - tt1, P.clock_probe[10] += tt3 - tt2, P.clock_probe[11] += 1;
        if (sno < 16) P.clock_probe[16 + sno] = tt1 - tt0, P.clock_probe[32 + sno] = tt2 - tt1, P.clock_probe[48 + sno] = tt3 - tt2;
        P.clock_probe[14] = tt3, P.clock_probe[15] = wall_clock64();
      }
#endif
      if (P.trace) {  // kernel-uniform; only the single-scan hook path sets it
        if (scan == 0 && lane == 0 && wit == 0) {
          float* t = P.trace + 12 * step;
          t[0] = ex; t[1] = ey; t[2] = eth;
          t[3] = acc.hd.x; t[4] = acc.h01; t[5] = acc.hr.x;
          t[6] = acc.h01; t[7] = acc.hd.y; t[8] = acc.hr.y;
          t[9] = acc.hr.x; t[10] = acc.hr.y; t[11] = acc.h22;
        }
        ++step;
      }
    }
    eth = normalize_angle(eth);                     // ScanMatcher.h:170
    affine_apply(L.worldTmap, ex, ey, pw0, pw1);    // getWorldCoordsPose, :186
    pw2 = eth;
  }
  if (lane == 0 && wit == 0) {
    P.out_pose[3 * scan + 0] = pw0;
    P.out_pose[3 * scan + 1] = pw1;
    P.out_pose[3 * scan + 2] = pw2;
    if (P.out_cov) {  // covMatrix = H of the last evaluation (ScanMatcher.h:184), column major
      float* c = P.out_cov + 9 * scan;
      c[0] = acc.hd.x; c[1] = acc.h01; c[2] = acc.hr.x;
      c[3] = acc.h01; c[4] = acc.hd.y; c[5] = acc.hr.y;
      c[6] = acc.hr.x; c[7] = acc.hr.y; c[8] = acc.h22;
    }
#ifdef HSM_TEAM_TIMELINE
    if (EXACT && P.clock_probe != nullptr && scan == 0) P.clock_probe[5] = wall_clock64();
#endif
    if (scan == 0) publish_done(P);
  }
}

// ---- one DENSE scan in the reference's summation order: producers ahead of the chain (round 5) ---------------------------------
// The exact-order team form above runs a 16 k-beam scan as rounds of 1024 beams on 16 wavefronts: all of them compute a round's
// products, then fifteen and a half of them wait while nine lanes add 1024 x 9 values (8.5 cycles per dependent v_add_f32: 3.6 us),
// then everybody gathers the next round's texels (~1-2 us with nothing else to do) -- 90 us per Gauss-Newton step on configs[4],
// of which the nine chains themselves are 16 384 x 8.5 cycles = 58 us.  This form takes the chain out of the team: wavefront 0
// only adds, wavefronts 1..15 only produce, one round AHEAD of it (two stage buffers, one workgroup barrier per round), so a
// round costs max(chain, production) = the chain.  Same products, same order of the additions: identical bits.  What is left
// per step is the floor of the literal chain -- n x 8.5 cycles -- plus one round of production before the first addition.
// (The block-scan form that would have removed the dependent chain itself is a measured negative: tools/study/exact_scan.h,
// profiles/r05/README.md.)
constexpr int kDenseProducers = 15;                  // wavefronts
constexpr int kDenseRound = 64 * kDenseProducers;    // beams per round: one per producer lane

template <int LAYOUT>
__global__ void __launch_bounds__(1024) gn_match_exact_dense_kernel(const MatchParams P) {
  constexpr int RL = kDenseRound;  // staged row length (a multiple of 16: exact_chain reads whole float4 groups)
  __shared__ __attribute__((aligned(16))) float stage[2][9 * (RL + kExactPad)];
  __shared__ float totals[9];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool chain_wave = wave == 0;
  const int ptid = (int)threadIdx.x - 64;  // producer lane 0 .. RL-1 (negative in the chain wavefront)
  const int scan = (int)blockIdx.x;
  int beg = 0, n = P.shared_n;
  if (P.offsets) {
    beg = P.offsets[scan];
    n = P.offsets[scan + 1] - beg;
  }
  float pw0, pw1, pw2;
  if (P.begin_world) {
    pw0 = P.begin_world[3 * scan + 0];
    pw1 = P.begin_world[3 * scan + 1];
    pw2 = P.begin_world[3 * scan + 2];
  } else {
    pw0 = P.begin_inline[0];
    pw1 = P.begin_inline[1];
    pw2 = P.begin_inline[2];
  }
  if (n == 0) {  // ScanMatcher.h:68,189
    if (threadIdx.x == 0) {
      P.out_pose[3 * scan + 0] = pw0;
      P.out_pose[3 * scan + 1] = pw1;
      P.out_pose[3 * scan + 2] = pw2;
      if (scan == 0) publish_done(P);
    }
    return;
  }
  const float2* __restrict__ pts = P.pts + beg;
  const int rounds = (n + RL - 1) / RL;
  // the chain is the critical path of every round: its wavefront issues ahead of the three producers that share its SIMD
  if (chain_wave) __builtin_amdgcn_s_setprio(3);
  auto endpoint_load = [&](int r) -> float2 {  // padding: an endpoint outside any map -> exact +-0 products (gn_match_kernel)
    const int i = r * RL + ptid;
    return (ptid >= 0 && i < n) ? pts[i] : make_float2(1.0e30f, 1.0e30f);
  };
  // a scan of one or two rounds (the node's 1081 beams) keeps its endpoints in registers across all levels and GN steps: one
  // dependent memory round trip per step -- the texel gather -- before the chain can start, instead of two
  const bool resident = rounds <= 2;  // (workgroup-uniform)
  float xq0x = 1.0e30f, xq0y = 1.0e30f, xq1x = 1.0e30f, xq1y = 1.0e30f;  // (scalars: a float2 selected through a lambda went to scratch)
  if (resident) {
    const float2 a = endpoint_load(0);
    xq0x = a.x, xq0y = a.y;
    if (rounds > 1) {
      const float2 b = endpoint_load(1);
      xq1x = b.x, xq1y = b.y;
    }
  }
  auto endpoint_of = [&](int r) -> float2 {
    if (!resident) return endpoint_load(r);
    return make_float2(r == 0 ? xq0x : xq1x, r == 0 ? xq0y : xq1y);
  };
  Acc9 acc;
  acc.zero();
  int step = 0;
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);
    eth = pw2;
    const float ps = L.pt_scale;
    const int gn_steps = L.gn_steps;
    const LevelRegs R = level_regs<LAYOUT>(L);
    for (int it = 0; it < gn_steps; ++it) {
      float sinRot, cosRot;
      sincos_f32(eth, sinRot, cosRot);
      const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
      // producers: the products of round r into stage[r & 1]; the endpoint of the round after it is requested meanwhile
      float2 q_next = endpoint_of(0);
      auto produce = [&](int r) {
        const float2 q = q_next;
        if (r + 1 < rounds) q_next = endpoint_of(r + 1);
        BeamRot rot;
        const BeamSample b = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{q.x * ps, q.y * ps}, rot);
        float pr[9];
        beam_products(b, rot, pr);
        exact_stage<RL>(pr, stage[r & 1], ptid);
      };
      if (!chain_wave) produce(0);
      float run = 0.0f;
      for (int r = 0; r < rounds; ++r) {  // workgroup-uniform trip count
        __syncthreads();  // round r is staged; the chain has left the other buffer
        if (chain_wave) {
          run = exact_chain<RL, false>(stage[r & 1], lane, run, min(RL, n - r * RL));
        } else if (r + 1 < rounds) {
          produce(r + 1);
        }
      }
      if (chain_wave && lane < 9) totals[lane] = run;
      __syncthreads();
      acc.d01 = f2{totals[0], totals[1]}; acc.d2 = totals[2];
      acc.hd = f2{totals[3], totals[4]}; acc.h22 = totals[5];
      acc.h01 = totals[6]; acc.hr = f2{totals[7], totals[8]};
      gn_solve_and_step(acc, ex, ey, eth);
      if (P.trace) {  // kernel-uniform; only the single-scan hook path sets it
        if (scan == 0 && threadIdx.x == 0) {
          float* t = P.trace + 12 * step;
          t[0] = ex; t[1] = ey; t[2] = eth;
          t[3] = acc.hd.x; t[4] = acc.h01; t[5] = acc.hr.x;
          t[6] = acc.h01; t[7] = acc.hd.y; t[8] = acc.hr.y;
          t[9] = acc.hr.x; t[10] = acc.hr.y; t[11] = acc.h22;
        }
        ++step;
      }
      // (the next step's first write of totals[] lies behind its rounds' barriers; stage[0] is rewritten by produce(0) of the
      // next step only after every wavefront has passed the barrier above, and the chain read it last in an earlier round)
    }
    eth = normalize_angle(eth);
    affine_apply(L.worldTmap, ex, ey, pw0, pw1);
    pw2 = eth;
  }
  if (threadIdx.x == 0) {
    P.out_pose[3 * scan + 0] = pw0;
    P.out_pose[3 * scan + 1] = pw1;
    P.out_pose[3 * scan + 2] = pw2;
    if (P.out_cov) {  // covMatrix = H of the last evaluation (ScanMatcher.h:184), column major
      float* c = P.out_cov + 9 * scan;
      c[0] = acc.hd.x; c[1] = acc.h01; c[2] = acc.hr.x;
      c[3] = acc.h01; c[4] = acc.hd.y; c[5] = acc.hr.y;
      c[6] = acc.hr.x; c[7] = acc.hr.y; c[8] = acc.h22;
    }
    if (scan == 0) publish_done(P);
  }
}

// ---- throughput form with a per-beam texel cache --------------------------------------------------
// From the third GN step on the estimate moves by a fraction of a cell, so most beams fall into the SAME
// map cell as in the step before (bench workload: 77 / 46 / 14 / 2 / 0.2 % of the lanes change cell in steps
// 2..6) and would gather the texel they already hold.  The gather, not the arithmetic, is what the texture
// path charges for -- one lane-request per cycle, whatever the pattern (profiles/r01/README.md) -- so this
// form keeps every beam's last texel and its byte offset in VGPRs (5 registers per beam) and gathers only in
// the lanes whose offset changed (exec-masked load; skipped by the whole wave when no lane changed).  To stay
// at 4 waves per SIMD (<= 128 VGPRs) the endpoints move from VGPRs to LDS: slot [wave][k][lane], written and
// read by the same lane only, so no barrier is ever needed (SPB * BPL * 512 B per workgroup).  Same
// arithmetic on the same texel values in the same order as gn_match_kernel: identical bits.
// One wave per scan, no trace.  The gathers run ONE beam ahead of their use (measured, profiles/r02/README.md: one
// ahead 57.1 us on the headline workload, two ahead 58.3 at 128 VGPRs, chunks of 3 without a pipeline 58.4).
// Compile-time switches of the measured variants (each A/B in profiles/r02/README.md):
#ifndef HSM_ASM_GATHER   // counted waits for the masked texel gathers (see locate())
#define HSM_ASM_GATHER 1
#endif
#ifndef HSM_LDS_AHEAD    // endpoint of beam k+2 read from LDS while beam k is consumed
#define HSM_LDS_AHEAD 1
#endif
#ifndef HSM_ZERO_VGPR
#define HSM_ZERO_VGPR 1
#endif

#ifndef HSM_GATHER_ALWAYS  // 1: every beam issues its (masked) gather with lane 0 enabled -- static load counts, no branches; measured neutral (49.5 / 95.7 / 133 us either way), off
#define HSM_GATHER_ALWAYS 0
#endif
#ifndef HSM_PEEL_FIRST   // first GN step takes the endpoints from their load registers (see the kernel)
#define HSM_PEEL_FIRST 1
#endif
#ifndef HSM_EP_AHEAD     // endpoint loads in flight ahead of the beam being located in the peeled step
#define HSM_EP_AHEAD 4
#endif

// Issue order of the loads of the peeled first step: endpoints E_0 .. E_d up front, the gather G_0, then per beam j
// the endpoint E_{j+1+d} and the gather G_{j+1}.  pos*[k] = index in that order; total = number of loads.
struct PeelSchedule {
  int posE[32], posG[32], total;
};
constexpr PeelSchedule peel_schedule(int bpl, int d) {
  PeelSchedule s{};
  int c = 0;
  for (int k = 0; k <= d && k < bpl; ++k) s.posE[k] = c++;
  s.posG[0] = c++;
  for (int j = 0; j < bpl; ++j) {
    if (j + 1 + d < bpl) s.posE[j + 1 + d] = c++;
    if (j + 1 < bpl) s.posG[j + 1] = c++;
  }
  s.total = c;
  return s;
}

// s_waitcnt vmcnt(n) for the inline-asm loads, ordered before every later use of `x` (the register the awaited load
// writes).  n is a compile-time constant after unrolling; anything above the cases waits for everything.
template <class T>
__device__ __forceinline__ void wait_vmcnt(int n, T& x) {
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" : "+v"(x) : : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(x) : : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" : "+v"(x) : : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(x) : : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" : "+v"(x) : : "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" : "+v"(x) : : "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" : "+v"(x) : : "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" : "+v"(x) : : "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" : "+v"(x) : : "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" : "+v"(x) : : "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" : "+v"(x) : : "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" : "+v"(x) : : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(x) : : "memory"); break;
  }
}

// a wave-uniform value moved to an SGPR
__device__ __forceinline__ float uniform_f32(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// 8-byte load from a 4-byte aligned address (two neighbouring cells of a row of the probability plane): one
// global_load_dwordx2 -- the hardware handles dword-aligned wide loads
struct __attribute__((packed, aligned(4))) CellPair {
  float a, b;
};

// WPS = 2 (experimental, HSM_CACHED_WPS2): TWO waves per scan, beam i in thread i mod 128 of the pair like
// gn_match_kernel<2,1> (identical bits), nine beams per lane = 92 VGPRs = five waves per SIMD: a 4096-scan launch is
// 8192 waves on 5120 slots, so late workgroups start as early ones finish.
//
// RELAXED (HSM_PARITY_RELAXED, opt-in): the same expressions with their multiply-add pairs CONTRACTED -- the rotation, the
// bilinear blend written as two lerps on the differences the gradient needs anyway, the blends of the differences, rotDeriv
// and the nine accumulations become v_fma_f32: 32 instead of 51 fp32 operations per beam (42 instead of 61 VALU
// instructions).  Per-beam values then differ from the reference's in the last bit or two (one rounding instead of two
// per pair); the mode's bar is north_star's tolerance (1e-4 m / 1e-4 rad on the pose), measured at full size against the
// reference (tests/test_gpu_full_size.py, bench.py), not bit-exactness of the terms.
template <int SPB, int BPL, int LAYOUT = kLayoutQuad, int WPS = 1, bool RELAXED = false>
__global__ void __launch_bounds__(64 * SPB * WPS, WPS > 1 ? 5 : 4) gn_match_cached_kernel(const MatchParams P) {
  static_assert(!RELAXED || (LAYOUT == kLayoutQuad && WPS == 1), "the tolerance mode exists for the throughput form only");
  static_assert(WPS == 1 || SPB == 1, "a pair of waves owns its workgroup (one barrier per GN step)");
  constexpr int T = 64 * WPS;  // lanes per scan
  __shared__ f2 lds_pts[SPB * WPS][BPL][64];
  __shared__ __attribute__((aligned(16))) float red[2][9][WPS < 4 ? 4 : WPS];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wit = __builtin_amdgcn_readfirstlane(wave % WPS);  // wave in team
  int red_buf = 0;
  // wave-uniform: kept in an SGPR (and with it the pose / covariance addresses, which live across the whole kernel)
  const int slot = __builtin_amdgcn_readfirstlane(xcd_block((int)blockIdx.x, (int)gridDim.x, P.xcd_chunk) * SPB + wave / WPS);
#if defined(HSM_EXPERIMENTS) && defined(HSM_EXP_TIMESTAMPS)
  const unsigned long long ts_entry = wall_clock64();
#endif
  if (slot >= P.batch) return;
  // (MatchParams::perm: the batch in Morton order of its start poses -- neighbours in the launch are neighbours in the map)
  const int scan = P.perm != nullptr ? __builtin_amdgcn_readfirstlane(P.perm[slot]) : slot;

  int beg = 0, n = P.shared_n;
  if (P.offsets) {
    beg = P.offsets[scan];
    n = P.offsets[scan + 1] - beg;
  }
  float pw0 = P.begin_world[3 * scan + 0], pw1 = P.begin_world[3 * scan + 1], pw2 = P.begin_world[3 * scan + 2];
  if (n == 0) {
    if (lane == 0 && wit == 0) {
      P.out_pose[3 * scan + 0] = pw0;
      P.out_pose[3 * scan + 1] = pw1;
      P.out_pose[3 * scan + 2] = pw2;
    }
    return;
  }
  const float2* __restrict__ pts = P.pts + beg;
  f2(*mine)[64] = lds_pts[wave];
  // Endpoint staging.  Every wave of a launch starts at the same time and needs its 8.6 KB of endpoints first: 35 MB for
  // 4096 scans, 6.5 us during which no wave has anything to compute if all endpoints are staged before the first GN
  // step (measured per wave with HSM_EXP_TIMESTAMPS, profiles/r02/README.md).  So the FIRST GN step of the first level
  // is peeled (kPeel): its beam k takes its endpoint straight from the register of a load issued kEpAhead beams
  // earlier and writes it to LDS for the later steps, so the arithmetic and the texel gathers of step one run while
  // the endpoints are still streaming in.  All loads of that step -- endpoints and (unmasked: every lane gathers in a
  // level's first step) texels -- are inline asm with counted waits: loads return in order, and peel_schedule() gives
  // the position of every load in the wave's issue order, hence how many younger loads may still be in flight when a
  // given one is needed.  Out-of-range lanes read the scan's last endpoint (exec stays full, so every load is issued
  // and the counts are static) and are replaced by the padding value.
  constexpr bool kPeel = LAYOUT == kLayoutQuad && HSM_ASM_GATHER && HSM_PEEL_FIRST;
  constexpr int kEpAhead = HSM_EP_AHEAD < BPL ? HSM_EP_AHEAD : BPL - 1;
  static_assert(BPL <= 31, "PeelSchedule holds 32 positions per load kind");
  constexpr PeelSchedule kSched = peel_schedule(BPL, kEpAhead);
  static_assert(kSched.total <= 2 * BPL, "one endpoint load and one gather per beam");
  const bool peel = kPeel && P.lv[P.first_level].gn_steps > 0;  // wave-uniform
  if (!peel) {
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
      const int i = lane + 64 * wit + k * T;
      const float2 q = i < n ? pts[i] : make_float2(1.0e30f, 1.0e30f);  // padding: see gn_match_kernel
      mine[k][lane] = f2{q.x, q.y};
    }
  }
#if defined(HSM_EXPERIMENTS) && defined(HSM_EXP_TIMESTAMPS)  // experiment (tools/exp_wave_timeline.py): per-wave start / end stamps of the 100 MHz clock
  const unsigned long long ts_begin = wall_clock64();
  const unsigned long long sc_begin = __builtin_readcyclecounter();  // shader clock (s_memtime)
#endif
  f4v tq[BPL];
  unsigned toff[BPL];
  Acc9 acc;
  acc.zero();
  float reg_scale = 1.0f;
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);
    eth = pw2;
    const float ps = L.pt_scale;
    const int gn_steps = L.gn_steps;
    const LevelRegs R = level_regs<LAYOUT>(L);
    const float ratio = ps / reg_scale;  // powers of two: exact (see gn_match_kernel)
    reg_scale = ps;
    const bool peel_here = peel && l == P.first_level;  // the peeled step stages the endpoints, scaled for this level
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
      if (!peel_here && ratio != 1.0f) mine[k][lane] *= f2{ratio, ratio};
      toff[k] = 0xffffffffu;  // never a texel offset (not a multiple of 16): every beam gathers in the first step
    }
    // byte offset of the all-zero texel, pinned in a VGPR (as an SGPR it costs a v_mov per beam in front of the select)
    unsigned zero_off = (unsigned)R.zero_index << (LAYOUT == kLayoutQuad ? 4 : 2);
#if HSM_ZERO_VGPR
    asm volatile("" : "+v"(zero_off));
#endif
    // one GN step; FIRST = the peeled step (compile-time)
    const bool wg_sync = __builtin_amdgcn_readfirstlane(P.wg_sync) != 0;
    auto gn_step = [&](auto FIRST, int it) {
      constexpr bool kFirst = decltype(FIRST)::value;
      // peeled step: the endpoint registers live only here.  The byte offset is made opaque so that its computation
      // stays at the load (hoisted out of the level loop, 17 offsets would sit in VGPRs for the whole kernel).
      f2 pq[kFirst ? BPL : 1];
      auto endpoint_issue = [&](int k) {
        int i = min((int)lane_id_now() + 64 * wit + T * k, n - 1);
        asm volatile("" : "+v"(i));
        const unsigned byte_off = (unsigned)i << 3;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(pq[kFirst ? k : 0]) : "v"(byte_off), "s"(pts) : "memory");
      };
      if (kFirst) {
        // clock probe, begin stamps (see the end of the kernel): here, at the top of the peeled step, no texel is live
        // yet -- the same block at the kernel's entry made the register allocator spill
        if (P.clock_probe != nullptr && scan == 0 && wit == 0 && lane_id_now() == 0) {
          P.clock_probe[0] = __builtin_readcyclecounter();
          P.clock_probe[1] = wall_clock64();
        }
#pragma unroll
        for (int k = 0; k <= kEpAhead; ++k) endpoint_issue(k);
      }
      rotate_wave_priority(it + l);
      float sinRot, cosRot;
      // (the tolerance mode keeps glibc's sincosf: with v_sin_f32 / v_cos_f32 or a 1-ulp fp32 polynomial it is 15-20 % instead of
      // 10-18 % faster than the fast mode, but one scan of the 32 768-scan sweep then lands 1.07e-4 m from the reference and
      // the worst case of the 3-level batch grows from 7.6e-6 to 9.5e-5 m -- profiles/r03/README.md)
      sincos_f32<true>(eth, sinRot, cosRot);
      acc.zero();
      // the step's pose and rotation are wave-uniform: held in SGPRs (4 VGPRs less in a kernel that has none to spare)
      const f2 o2 = step_origin(ex, ey);
      const f2 e2 = f2{uniform_f32(o2.x), uniform_f32(o2.y)};
      const f2 cs = f2{uniform_f32(cosRot), uniform_f32(sinRot)}, sc = f2{cs.y, cs.x};
      // "locate" a beam: rotate, bounds test, cell offset, fractions, and -- only in the lanes whose cell changed since
      // the previous step -- the gather of its texel straight into the beam's cache registers
      auto locate = [&](int k, f2 p, BeamRot& r, float& fx, float& fy) -> unsigned long long {
        if (RELAXED) {
          r.r.x = __builtin_fmaf(cs.x, p.x, -(sc.x * p.y));
          r.r.y = __builtin_fmaf(cs.y, p.x, sc.y * p.y);
        } else {
          r.r.x = cs.x * p.x - sc.x * p.y;
          r.r.y = cs.y * p.x + sc.y * p.y;
        }
        const CellCoord q = cell_coord(R, f2{e2.x + r.r.x, e2.y + r.r.y});
        fx = q.fx;
        fy = q.fy;
        unsigned idx = LAYOUT == kLayoutQuad ? quad_index(q.ix, q.iy, R.tiles_x, R.sx) : __umul24(q.iy, (unsigned)R.sx) + q.ix;
        asm volatile("" : "+v"(idx));  // computed unconditionally: a select below, not a branch
        const unsigned off = q.oob ? zero_off : idx << (LAYOUT == kLayoutQuad ? 4 : 2);
        if (kFirst) {  // a level's first step: every lane gathers (toff[] holds no offset yet)
          asm volatile("global_load_dwordx4 %[t], %[o], %[b]" : [t] "=v"(tq[k]) : [o] "v"(off), [b] "s"(R.quad) : "memory");
          toff[k] = off;
          return ~0ull;
        }
        if (LAYOUT == kLayoutQuad && HSM_ASM_GATHER) {
          // The masked gather as ONE instruction sequence under the wave's own control.  The compiler's form of
          // `if (off != toff[k]) load` waits with vmcnt(0) before the PREVIOUS beam is consumed -- it cannot count
          // loads that sit behind a branch -- which puts the gather just issued on the critical path and defeats the
          // software pipeline below.  Here the wave records whether the gather was issued (the mask of the lanes that
          // moved, wave-uniform) and texel_ready() waits for exactly the load it needs.  No C++ control flow: the
          // compiler sees straight-line code and keeps tq[k] where it is (it does not know about the asynchronous
          // write; texel_ready(k) is ordered before every read of tq[k] through its "+v" operand).
          unsigned long long moved, saved;
#if HSM_GATHER_ALWAYS
          // lane 0 always re-reads its texel (same value: the map does not change under the kernel), so every beam
          // issues exactly ONE load and the waits are static -- no branch around the load, none around the wait
          asm volatile(
              "v_cmp_ne_u32 vcc, %[o], %[to]\n\t"
              "s_or_b32 vcc_lo, vcc_lo, 1\n\t"
              "s_and_saveexec_b64 %[sv], vcc\n\t"
              "global_load_dwordx4 %[t], %[o], %[b]\n\t"
              "v_mov_b32 %[to], %[o]\n\t"
              "s_mov_b64 exec, %[sv]"
              : [t] "+v"(tq[k]), [to] "+v"(toff[k]), [sv] "=&s"(saved)
              : [o] "v"(off), [b] "s"(R.quad)
              : "vcc", "scc", "memory");
          moved = ~0ull;
#else
          asm volatile(
              "v_cmp_ne_u32 vcc, %[o], %[to]\n\t"
              "s_mov_b64 %[mv], vcc\n\t"
              "s_and_saveexec_b64 %[sv], vcc\n\t"
              "s_cbranch_execz 1f\n\t"
              "global_load_dwordx4 %[t], %[o], %[b]\n\t"
              "v_mov_b32 %[to], %[o]\n\t"
              "1:\n\t"
              "s_mov_b64 exec, %[sv]"
              : [t] "+v"(tq[k]), [to] "+v"(toff[k]), [sv] "=&s"(saved), [mv] "=&s"(moved)
              : [o] "v"(off), [b] "s"(R.quad)
              : "vcc", "scc", "memory");
#endif
          return moved;
        }
        if (off != toff[k]) {
          if (LAYOUT == kLayoutQuad) {
            tq[k] = *reinterpret_cast<const f4v*>(reinterpret_cast<const char*>(R.quad) + (size_t)off);
          } else {
            // the probability plane itself (4 B per cell: a quarter of the texel plane's footprint, so far more of
            // a map that outgrows the L2 stays in it): rows iy and iy + 1, two cells each
            const char* row = reinterpret_cast<const char*>(R.prob) + (size_t)off;
            const CellPair lo = *reinterpret_cast<const CellPair*>(row);
            const CellPair hi = *reinterpret_cast<const CellPair*>(row + ((size_t)R.sx << 2));
            tq[k] = f4v{lo.a, lo.b, hi.a, hi.b};
          }
          toff[k] = off;
        }
        return 0ull;
      };
      // beam k's texel has landed.  `next_moved` = the mask returned by the locate() of the only gather that may have
      // been issued after beam k's (wave-uniform): loads return in order, so with it in flight vmcnt(1) is enough.
      auto texel_ready = [&](int k, unsigned long long next_moved, bool has_next) {
        if (!(LAYOUT == kLayoutQuad && HSM_ASM_GATHER)) return;
        if (kFirst) {  // static schedule: everything issued after beam k's gather may still be in flight
          wait_vmcnt((has_next ? kSched.posG[k + 1] + 1 : kSched.total) - kSched.posG[k] - 1, tq[k]);
        } else if (has_next && HSM_GATHER_ALWAYS) {
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(tq[k]) : : "memory");
        } else if (has_next) {
          asm volatile(
              "s_cmp_eq_u64 %[m], 0\n\t"
              "s_cbranch_scc1 1f\n\t"
              "s_waitcnt vmcnt(1)\n\t"
              "s_branch 2f\n\t"
              "1:\n\t"
              "s_waitcnt vmcnt(0)\n\t"
              "2:"
              : "+v"(tq[k])
              : [m] "s"(next_moved)
              : "scc", "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(tq[k]) : : "memory");
        }
      };
      // peeled step: endpoint k from its load register (padding lanes replaced), scaled for this level, into LDS
      auto endpoint_take = [&](int k) -> f2 {
        wait_vmcnt(kSched.posG[k] - kSched.posE[k] - 1, pq[k]);  // beam k's gather is the next load in issue order
        const bool pad = (int)lane_id_now() + 64 * wit + T * k >= n;
        const f2 p = f2{(pad ? 1.0e30f : pq[k].x) * ps, (pad ? 1.0e30f : pq[k].y) * ps};
        mine[k][lane] = p;
        return p;
      };
      auto consume = [&](int k, const BeamRot& r, float fx, float fy) {
        if (RELAXED) {
          // OccGridMapUtil.h:332-346 and :80-97 with contracted multiply-adds: the blend as two lerps on the x differences
          // (i0*(1-fx) + i1*fx == i0 - fx*(i0-i1)), then one on the rows; accumulations as v_fma_f32
          const float i0 = tq[k].x, i1 = tq[k].y, i2 = tq[k].z, i3 = tq[k].w;
          const float xFacInv = 1.0f - fx, yFacInv = 1.0f - fy;
          const float dx1 = i0 - i1, dx2 = i2 - i3, dy1 = i0 - i2, dy2 = i1 - i3;
          const float t0 = __builtin_fmaf(-fx, dx1, i0), t1 = __builtin_fmaf(-fx, dx2, i2);
          const float M = __builtin_fmaf(fy, t1 - t0, t0);
          const float Gx = __builtin_fmaf(dx2, fx, dx1 * xFacInv);  // -gx
          const float Gy = __builtin_fmaf(dy2, fy, dy1 * yFacInv);  // -gy
          const float funVal = 1.0f - M;
          const float rotDeriv = __builtin_fmaf(r.r.y, Gx, -(r.r.x * Gy));
          acc.d01.x = __builtin_fmaf(-Gx, funVal, acc.d01.x);
          acc.d01.y = __builtin_fmaf(-Gy, funVal, acc.d01.y);
          acc.d2 = __builtin_fmaf(rotDeriv, funVal, acc.d2);
          acc.hd.x = __builtin_fmaf(Gx, Gx, acc.hd.x);
          acc.hd.y = __builtin_fmaf(Gy, Gy, acc.hd.y);
          acc.h22 = __builtin_fmaf(rotDeriv, rotDeriv, acc.h22);
          acc.h01 = __builtin_fmaf(Gx, Gy, acc.h01);
          acc.hr.x = __builtin_fmaf(-Gx, rotDeriv, acc.hr.x);
          acc.hr.y = __builtin_fmaf(-Gy, rotDeriv, acc.hr.y);
          return;
        }
        BeamSample b;
        b.X = f2{1.0f - fx, fx};
        b.Y = f2{1.0f - fy, fy};
        b.lo = f2{tq[k].x, tq[k].y};
        b.hi = f2{tq[k].z, tq[k].w};
        beam_finish(b, r, acc);
      };
      // Software pipeline: the texel gather of beam k+1 is ISSUED before beam k is consumed, so a gather has the
      // arithmetic of a whole beam (and the other waves' share of the SIMD) to land in.  The gathers write the beams'
      // own cache registers, so the only extra state in flight is the next beam's (rot, fx, fy); the endpoint of beam
      // k+2 is read from LDS before beam k+1 is located (HSM_LDS_AHEAD), so the LDS latency is off the chain too.
      // Same arithmetic in the same beam order: identical bits.
      {
        BeamRot rc, rn;
        float fxc, fyc, fxn = 0.0f, fyn = 0.0f;
        f2 p_next = f2{0.0f, 0.0f};
        unsigned long long next_moved = 0ull;
        if (kFirst) {
          locate(0, endpoint_take(0), rc, fxc, fyc);
        } else {
          p_next = mine[BPL > 1 ? 1 : 0][lane];
          locate(0, mine[0][lane], rc, fxc, fyc);
        }
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
          if (kFirst) {
            if (k + 1 + kEpAhead < BPL) endpoint_issue(k + 1 + kEpAhead);
            if (k + 1 < BPL) next_moved = locate(k + 1, endpoint_take(k + 1), rn, fxn, fyn);
          } else {
            const f2 p_cur = p_next;
            if (HSM_LDS_AHEAD && k + 2 < BPL) p_next = mine[k + 2][lane];
            if (k + 1 < BPL) next_moved = locate(k + 1, HSM_LDS_AHEAD ? p_cur : mine[k + 1][lane], rn, fxn, fyn);
          }
          // large maps (MatchParams::wg_sync): the four waves of a workgroup -- consecutive scans, whose beam k ends in
          // the same or neighbouring cells -- locate beam k together, so the texel lines one of them pulls in are still
          // in the CU's L1 when the others ask (4096^2 pyramid: 137 -> 133 us; nothing on maps whose touched region the
          // L2s hold).  Waves that have left the kernel (empty scan, batch tail) do not count at s_barrier.
          if (wg_sync) asm volatile("s_barrier" ::: "memory");
          texel_ready(k, next_moved, k + 1 < BPL);
          consume(k, rc, fxc, fyc);
          asm volatile(""
                       : "+v"(acc.d01), "+v"(acc.d2), "+v"(acc.hd), "+v"(acc.h22), "+v"(acc.h01), "+v"(acc.hr)
                       :
                       : "memory");
          rc = rn;
          fxc = fxn;
          fyc = fyn;
        }
      }
      // a scan longer than the 64 * BPL cached beams (BPL comes from a host-side length HINT): the rest streams
      // from memory like gn_match_kernel's loop, in the same per-lane order (wave-uniform trip count)
      for (int i = T * BPL + (n > T * BPL ? lane_id_now() + 64 * wit : 0); i < n; i += T) {
        const float2 p = pts[i];
        BeamRot r;
        const BeamSample b = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p.x * ps, p.y * ps}, r);
        beam_finish(b, r, acc);
      }
      if (WPS > 1) {
        team_allreduce9<WPS>(acc, red, red_buf, wit, lane_id_now());
        red_buf ^= 1;
      } else {
        wave_allreduce9(acc);
      }
      gn_solve_and_step(acc, ex, ey, eth);
    };
    int it = 0;
    if (kPeel && peel_here) {
      gn_step(std::true_type{}, 0);
      it = 1;
    }
    for (; it < gn_steps; ++it) gn_step(std::false_type{}, it);
    eth = normalize_angle<true>(eth);
    affine_apply(L.worldTmap, ex, ey, pw0, pw1);
    pw2 = eth;
  }
  if (lane_id_now() == 0 && wit == 0) {
    P.out_pose[3 * scan + 0] = pw0;
    P.out_pose[3 * scan + 1] = pw1;
    P.out_pose[3 * scan + 2] = pw2;
    if (P.out_cov) {
      float* c = P.out_cov + 9 * scan;
#if defined(HSM_EXPERIMENTS) && defined(HSM_EXP_TIMESTAMPS)  // the stamps and the wave's placement overwrite the covariance
      const unsigned long long ts_end = wall_clock64();
      const unsigned long long sc_end = __builtin_readcyclecounter();
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned* u = reinterpret_cast<unsigned*>(c);
      u[0] = (unsigned)ts_begin; u[1] = (unsigned)(ts_begin >> 32); u[2] = (unsigned)ts_end; u[3] = (unsigned)(ts_end >> 32);
      u[4] = hwid; u[5] = xcc; u[6] = blockIdx.x; u[7] = (unsigned)(ts_begin - ts_entry); u[8] = (unsigned)(sc_end - sc_begin);
#else
      c[0] = acc.hd.x; c[1] = acc.h01; c[2] = acc.hr.x;
      c[3] = acc.h01; c[4] = acc.hd.y; c[5] = acc.hr.y;
      c[6] = acc.hr.x; c[7] = acc.hr.y; c[8] = acc.h22;
#endif
    }
  }
  // clock probe (bench.py: what clock does the kernel actually get?): the wave of scan 0 stamps the shader-clock counter
  // (s_memtime) and the 100 MHz wall clock at the top of its first GN step and here, as its last act; the ratio of the
  // two differences is the clock it ran at.  (Stamps of different launches cannot be compared: the wave lands on
  // different CUs, whose shader-clock counters are not aligned.)
  if (P.clock_probe != nullptr && scan == 0 && wit == 0 && lane_id_now() == 0) {
    P.clock_probe[2] = __builtin_readcyclecounter();
    P.clock_probe[3] = wall_clock64();
  }
}

// ---- HSM_PARITY_EXACT for batches: wave-specialised -----------------------------------------------------------------
// In exact_round() nine lanes of a wavefront run the nine sequential chains while the other 55 idle: 64 dependent adds
// (+ the LDS traffic) per 64 beams, about as much issue time as the beam arithmetic itself.  A batch has many scans, so
// this form lets ONE consumer wavefront run the chains of SEVEN scans side by side -- lane 9 j + t adds term t of scan
// j -- while seven producer wavefronts (one scan each) compute the products: the chain cost per scan drops 7x and it
// overlaps the producers' arithmetic.  Per round of 64 beams every producer writes its 9 x 64 products into its slice
// of a double-buffered LDS stage and the workgroup meets at one barrier; the consumer then sums round r while the
// producers already compute round r + 1 (buffer reuse is safe: the consumer reaches barrier r + 1 only after chain r).
// After the last round the consumer publishes the 7 x 9 totals; every producer picks up its nine and solves.  The
// round count is the workgroup's longest scan (shorter scans pad with +-0 products, which leave a sum unchanged).
// Summation order per scan: beam 0 .. n-1, one fp32 chain per term -- the reference's (OccGridMapUtil.h:76-98), so the
// results are bit-identical to gn_match_kernel<..., EXACT> and to the reference.
// Workgroup shape <NPROD producers, NCONS consumers>: a consumer wavefront runs the chains of NPROD / NCONS <= 7 scans.
// <7, 1> fills the consumer (63 chain lanes); <8, 2> makes a 4096-scan batch 512 workgroups = exactly two per CU --
// with <7, 1> it is 586 workgroups, 74 of the 256 CUs get three of them and the launch lasts as long as those.
constexpr int kExactScans = 7;  // <7, 1>: producers per workgroup; 7 x 9 = 63 chain lanes in the consumer wavefront

#ifndef HSM_EXACT_DEEP  // <8, 2> shape: two texel gathers in flight per producer (needs > 64 VGPRs: 5 waves per SIMD)
#define HSM_EXACT_DEEP 1
#endif
template <int LAYOUT, int NPROD = kExactScans, int NCONS = 1>
__global__ void __launch_bounds__(64 * (NPROD + NCONS), (NPROD == 8 && HSM_EXACT_DEEP) ? 5 : 8)
gn_match_exact_batch_kernel(const MatchParams P) {
  constexpr bool kDeep = NPROD == 8 && HSM_EXACT_DEEP;
  constexpr int kExactScans = NPROD;  // shadows the namespace constant: scans (= producer wavefronts) per workgroup
  constexpr int SPC = NPROD / NCONS;  // scans per consumer wavefront
  static_assert(NPROD % NCONS == 0 && SPC <= 7, "a consumer wavefront has 64 lanes for 9 chains per scan");
  constexpr int ROW = 64 + kExactPad;
  __shared__ float stage[2][kExactScans][9][ROW];
  __shared__ float tot[kExactScans][9];
  __shared__ int nmax_s;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool consumer = wave >= kExactScans;
  const int scan = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kExactScans + wave);
  const bool active = !consumer && scan < P.batch;
  int beg = 0, n = 0;
  float pw0 = 0.0f, pw1 = 0.0f, pw2 = 0.0f;
  if (active) {
    n = P.shared_n;
    if (P.offsets) {
      beg = P.offsets[scan];
      n = P.offsets[scan + 1] - beg;
    }
    pw0 = P.begin_world[3 * scan + 0];
    pw1 = P.begin_world[3 * scan + 1];
    pw2 = P.begin_world[3 * scan + 2];
  }
  const float b0 = pw0, b1 = pw1, b2 = pw2;  // an empty scan passes its start estimate through untouched (ScanMatcher.h:68,189)
  if (threadIdx.x == 0) nmax_s = 0;
  __syncthreads();
  if (lane == 0 && n > 0) atomicMax(&nmax_s, n);
  __syncthreads();
  const int rounds = (nmax_s + 63) >> 6;  // workgroup-uniform
  if (rounds == 0) {                      // nothing but empty scans
    if (active && lane == 0) {
      P.out_pose[3 * scan + 0] = b0;
      P.out_pose[3 * scan + 1] = b1;
      P.out_pose[3 * scan + 2] = b2;
    }
    return;
  }
  const float2* __restrict__ pts = P.pts + beg;
  Acc9 acc;
  acc.zero();
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);
    eth = pw2;
    const float ps = L.pt_scale;
    const int gn_steps = L.gn_steps;
    const LevelRegs R = level_regs<LAYOUT>(L);
    for (int it = 0; it < gn_steps; ++it) {
      if (consumer) {
        float run = 0.0f;
        const int jl = lane / 9, t = lane - 9 * jl;  // lanes >= 9 * SPC idle
        const int j = (wave - kExactScans) * SPC + (jl < SPC ? jl : 0);
        for (int r = 0; r < rounds; ++r) {
          __syncthreads();  // the producers' products of round r are in stage[r & 1]
          if (lane < 9 * SPC) {
            const float* row = &stage[r & 1][j][t][0];
#pragma unroll 4
            for (int q = 0; q < 64; q += 4) {
              const float4 v = *reinterpret_cast<const float4*>(row + q);
              run += v.x;
              run += v.y;
              run += v.z;
              run += v.w;
            }
          }
        }
        if (lane < 9 * SPC) tot[j][t] = run;
        __syncthreads();  // totals published
      } else {
        float sinRot, cosRot;
        sincos_f32(eth, sinRot, cosRot);
        const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
        // two-deep software pipeline: the endpoint of round r + 2 and the texel of round r + 1 are in flight while
        // the products of round r are computed, so the barrier of a round does not wait for a memory round trip
        const float2 pad = make_float2(1.0e30f, 1.0e30f);  // padding: exact +-0 products
        float2 p_next = lane < n ? pts[lane] : pad;
        BeamRot rot_next, rot_next2;
        BeamSample b_next = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p_next.x * ps, p_next.y * ps}, rot_next), b_next2 = b_next;
        p_next = 64 + lane < n ? pts[64 + lane] : pad;
        if (kDeep) {  // three-deep: texels of rounds r + 1 and r + 2 and the endpoint of round r + 3 in flight
          b_next2 = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p_next.x * ps, p_next.y * ps}, rot_next2);
          p_next = 128 + lane < n ? pts[128 + lane] : pad;
        }
        for (int r = 0; r < rounds; ++r) {
          const BeamSample b = b_next;
          const BeamRot rot = rot_next;
          if (kDeep) {
            b_next = b_next2;
            rot_next = rot_next2;
            if (r + 2 < rounds) {
              b_next2 = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p_next.x * ps, p_next.y * ps}, rot_next2);
              const int i3 = ((r + 3) << 6) + lane;
              p_next = i3 < n ? pts[i3] : pad;
            }
          } else if (r + 1 < rounds) {
            b_next = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p_next.x * ps, p_next.y * ps}, rot_next);
            const int i2 = ((r + 2) << 6) + lane;
            p_next = i2 < n ? pts[i2] : pad;
          }
          float pr[9];
          beam_products(b, rot, pr);
          float* st = &stage[r & 1][wave][0][lane];
#pragma unroll
          for (int t = 0; t < 9; ++t) st[t * ROW] = pr[t];
          __syncthreads();
        }
        __syncthreads();  // the consumer has published the totals
        const float* tt = &tot[wave][0];
        acc.d01 = f2{tt[0], tt[1]}; acc.d2 = tt[2];
        acc.hd = f2{tt[3], tt[4]}; acc.h22 = tt[5];
        acc.h01 = tt[6]; acc.hr = f2{tt[7], tt[8]};
        gn_solve_and_step(acc, ex, ey, eth);
      }
    }
    if (!consumer) {
      eth = normalize_angle(eth);
      affine_apply(L.worldTmap, ex, ey, pw0, pw1);
      pw2 = eth;
    }
  }
  if (active && lane == 0) {
    const bool empty = n == 0;
    P.out_pose[3 * scan + 0] = empty ? b0 : pw0;
    P.out_pose[3 * scan + 1] = empty ? b1 : pw1;
    P.out_pose[3 * scan + 2] = empty ? b2 : pw2;
    if (P.out_cov && !empty) {
      float* c = P.out_cov + 9 * scan;
      c[0] = acc.hd.x; c[1] = acc.h01; c[2] = acc.hr.x;
      c[3] = acc.h01; c[4] = acc.hd.y; c[5] = acc.hr.y;
      c[6] = acc.hr.x; c[7] = acc.hr.y; c[8] = acc.h22;
    }
  }
}

// ---- one DENSE scan on many CUs --------------------------------------------------------------------
// gn_match_kernel keeps a scan inside one workgroup (<= 16 waves on ONE CU): right for batches, but a
// single 16k-beam scan then uses 1/256 of the chip (11 us per GN step).  This variant spreads the beams of
// ONE scan over K <= 64 workgroups of 256 lanes and keeps the 14-step chain inside one cooperative launch:
// per GN step every workgroup reduces its beams to 9 partials, publishes them, the grid synchronises
// (cooperative groups: all K workgroups are co-resident by construction, the runtime refuses the launch
// otherwise), and every workgroup sums the K partials in the same fixed order -- wave 0, lane l takes
// workgroup l, then the DPP/permlane tree -- so all of them take the identical GN step.  Partials are double
// buffered by step parity: one grid sync per step.
// Grid barrier of the cooperative matcher: one monotonically increasing device-memory counter (the host passes
// its value at launch, so it is never reset), one agent-scope release increment per workgroup and an acquire
// spin by thread 0 -- 2-3 us per GN step cheaper than cooperative_groups' grid.sync() (HSM_COOP_BARRIER=0
// selects that one).  All K workgroups are co-resident: the launch is still a cooperative launch.
#ifndef HSM_COOP_BARRIER
#define HSM_COOP_BARRIER 1
#endif
// TAGGED (default since round 3; env HSM_COOP_TAGGED=0 selects the counter barrier at run time): no grid barrier at all --
// tagged 16-byte records, see the kernel.  That exchange relies on a 16-byte sc0 sc1 store being observed untorn, which is
// documented behaviour of gfx942 / gfx950 only:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "gn_match_coop_kernel's tagged exchange is written for gfx942 / gfx950"
#endif
__device__ __forceinline__ void coop_barrier(unsigned* counter, unsigned target) {
  __syncthreads();  // the workgroup's partials are written
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // relaxed polls (each acquire load would invalidate the caches), ONE acquire fence once everybody arrived
    while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
      __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int LAYOUT, bool TAGGED = true>
__global__ void __launch_bounds__(256) gn_match_coop_kernel(const MatchParams P, float* __restrict__ partials,
                                                            unsigned* __restrict__ bar_counter, unsigned bar_base) {
#if !HSM_COOP_BARRIER
  namespace cg = cooperative_groups;
  cg::grid_group grid = cg::this_grid();
#endif
  __shared__ float red[4][9];
  __shared__ float tot[9];
  __shared__ int gave_up;  // the exchange timed out in this workgroup: leave (ordered by the barriers of the step)
  if (threadIdx.x == 0) gave_up = 0;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int K = (int)gridDim.x;
  const int n = P.shared_n;
  const float2* __restrict__ pts = P.pts;
  float pw0 = P.begin_inline[0], pw1 = P.begin_inline[1], pw2 = P.begin_inline[2];
  const int g0 = blockIdx.x * 256 + threadIdx.x, stride = K * 256;
  Acc9 acc;
  acc.zero();
  int step = 0;
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);
    eth = pw2;
    const float ps = L.pt_scale;
    const int gn_steps = L.gn_steps;
    const LevelRegs R = level_regs<LAYOUT>(L);
    for (int it = 0; it < gn_steps; ++it, ++step) {
      float sinRot, cosRot;
      sincos_f32(eth, sinRot, cosRot);
      acc.zero();
      const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
      for (int i = g0; i < n; i += stride) {
        const float2 p = pts[i];
        BeamRot r;
        const BeamSample b = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p.x * ps, p.y * ps}, r);
        beam_finish(b, r, acc);
      }
      // workgroup partial: wave all-reduce, 4 waves through LDS (fixed order)
      wave_allreduce9(acc);
      if (lane == 0) {
        float* r = red[wave];
        r[0] = acc.d01.x; r[1] = acc.d01.y; r[2] = acc.d2;
        r[3] = acc.hd.x; r[4] = acc.hd.y; r[5] = acc.h22;
        r[6] = acc.h01; r[7] = acc.hr.x; r[8] = acc.hr.y;
      }
      __syncthreads();
      if constexpr (TAGGED) {
      // Exchange WITHOUT a barrier: every workgroup publishes its nine partials as three self-describing 16-byte granules
      // {p, p, p, tag} (tag = the step's global sequence number) with device-coherent stores, and every workgroup polls the K
      // records of the step until all their granules carry the tag.  A 16-byte sc0 sc1 store is observed untorn, sc1 loads are
      // served by memory-side coherent L2 -- no release write-back, no acquire invalidate (1.7 us each on this machine), no
      // counter round trip.  Two record buffers by step parity: a workgroup overwrites buffer s & 1 for step s + 2 only after
      // it has read every workgroup's step-(s + 1) record, which that workgroup published after it finished reading step s.
      const unsigned tag = bar_base + (unsigned)step + 1u;
      f4v* const recs = reinterpret_cast<f4v*>(partials) + (size_t)(step & 1) * 64 * 3;
      if (threadIdx.x < 3 && P.coop_mute_block != (int)blockIdx.x + 1) {
        const int t0 = 3 * (int)threadIdx.x;
        f4v g;
        g.x = ((red[0][t0] + red[1][t0]) + red[2][t0]) + red[3][t0];
        g.y = ((red[0][t0 + 1] + red[1][t0 + 1]) + red[2][t0 + 1]) + red[3][t0 + 1];
        g.z = ((red[0][t0 + 2] + red[1][t0 + 2]) + red[2][t0 + 2]) + red[3][t0 + 2];
        g.w = __uint_as_float(tag);
        f4v* dst = recs + 3 * blockIdx.x + threadIdx.x;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(g) : "memory");
      }
      if (wave == 0) {
        Acc9 t;
        t.zero();
        const f4v* src = recs + 3 * (lane < K ? lane : 0);
        f4v g0, g1, g2;
        for (int spin = 0;; ++spin) {
          asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\t"
                       "global_load_dwordx4 %1, %3, off offset:16 sc0 sc1\n\t"
                       "global_load_dwordx4 %2, %3, off offset:32 sc0 sc1\n\t"
                       "s_waitcnt vmcnt(0)"
                       : "=&v"(g0), "=&v"(g1), "=&v"(g2) : "v"(src) : "memory");
          const bool ok = lane >= K || (__float_as_uint(g0.w) == tag && __float_as_uint(g1.w) == tag && __float_as_uint(g2.w) == tag);
          if (__ballot(!ok) == 0ull) break;
          if (spin > (1 << 22)) {
            // bounded: a lost workgroup must not hang the device.  The records are then stale or partial, and so is every
            // step from here on: say so where the host looks (match_single turns it into an error return) -- ordered
            // before this workgroup's next record, so that whoever consumes the wrong sums also finds the word
            if (lane == 0 && P.err_flag) {
              __hip_atomic_store(P.err_flag, P.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __threadfence_system();
            }
            // ... and leave: every other workgroup times out on this one's missing records in turn, so the launch ends after
            // ONE bounded wait instead of one per remaining GN step (the host re-runs the scan on the one-workgroup matcher)
            if (lane == 0) gave_up = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane < K) {
          t.d01 = f2{g0.x, g0.y}; t.d2 = g0.z;
          t.hd = f2{g1.x, g1.y}; t.h22 = g1.z;
          t.h01 = g2.x; t.hr = f2{g2.y, g2.z};
        }
        wave_allreduce9(t);
        if (lane == 0) {
          tot[0] = t.d01.x; tot[1] = t.d01.y; tot[2] = t.d2;
          tot[3] = t.hd.x; tot[4] = t.hd.y; tot[5] = t.h22;
          tot[6] = t.h01; tot[7] = t.hr.x; tot[8] = t.hr.y;
        }
      }
      } else {
      float* mine = partials + ((size_t)(step & 1) * K + blockIdx.x) * 9;
      if (threadIdx.x < 9) mine[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
#if HSM_COOP_BARRIER
      coop_barrier(bar_counter, bar_base + (unsigned)K * (unsigned)(step + 1));
#else
      grid.sync();
#endif
      // every workgroup: the same K partials in the same order
      if (wave == 0) {
        const float* src = partials + ((size_t)(step & 1) * K + lane) * 9;
        Acc9 t;
        t.zero();
        if (lane < K) {
          t.d01 = f2{src[0], src[1]}; t.d2 = src[2];
          t.hd = f2{src[3], src[4]}; t.h22 = src[5];
          t.h01 = src[6]; t.hr = f2{src[7], src[8]};
        }
        wave_allreduce9(t);
        if (lane == 0) {
          tot[0] = t.d01.x; tot[1] = t.d01.y; tot[2] = t.d2;
          tot[3] = t.hd.x; tot[4] = t.hd.y; tot[5] = t.h22;
          tot[6] = t.h01; tot[7] = t.hr.x; tot[8] = t.hr.y;
        }
      }
      }
      __syncthreads();
      if (TAGGED && gave_up) return;  // (workgroup-uniform)
      acc.d01 = f2{tot[0], tot[1]}; acc.d2 = tot[2];
      acc.hd = f2{tot[3], tot[4]}; acc.h22 = tot[5];
      acc.h01 = tot[6]; acc.hr = f2{tot[7], tot[8]};
      gn_solve_and_step(acc, ex, ey, eth);
      if (P.trace && blockIdx.x == 0 && threadIdx.x == 0) {
        float* t = P.trace + 12 * step;
        t[0] = ex; t[1] = ey; t[2] = eth;
        t[3] = acc.hd.x; t[4] = acc.h01; t[5] = acc.hr.x;
        t[6] = acc.h01; t[7] = acc.hd.y; t[8] = acc.hr.y;
        t[9] = acc.hr.x; t[10] = acc.hr.y; t[11] = acc.h22;
      }
      __syncthreads();  // tot[] / red[] are rewritten in the next step
    }
    eth = normalize_angle(eth);
    affine_apply(L.worldTmap, ex, ey, pw0, pw1);
    pw2 = eth;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    P.out_pose[0] = pw0;
    P.out_pose[1] = pw1;
    P.out_pose[2] = pw2;
    if (P.out_cov) {
      float* c = P.out_cov;
      c[0] = acc.hd.x; c[1] = acc.h01; c[2] = acc.hr.x;
      c[3] = acc.h01; c[4] = acc.hd.y; c[5] = acc.hr.y;
      c[6] = acc.hr.x; c[7] = acc.hr.y; c[8] = acc.h22;
    }
    publish_done(P);
  }
}

// ---- parity / debug kernels: one evaluation at a given map-frame pose ------------
// H, dTr of one getCompleteHessianDerivs call (same device functions as the matcher)
template <int LAYOUT, bool EXACT = false>
__global__ void __launch_bounds__(1024) gn_eval_kernel(const LevelView L, const float2* __restrict__ pts,
                                                      int n, float ex, float ey, float eth,
                                                      float* out12 /* H[9] col-major, dTr[3] */) {
  __shared__ __attribute__((aligned(16))) float red[2][9][16];
  __shared__ float stage[EXACT ? 9 * (1024 + kExactPad) : 1];
  const int lane = threadIdx.x & 63;
  const int wit = threadIdx.x >> 6;
  float sinRot, cosRot;
  sincos_f32(eth, sinRot, cosRot);
  Acc9 acc;
  acc.zero();
  const LevelRegs R = level_regs<LAYOUT>(L);
  if (EXACT) {
    const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
    float run = 0.0f;
    for (int base = 0; base < n; base += 1024) {
      const int i = base + (int)threadIdx.x;
      const float2 p = i < n ? pts[i] : make_float2(1.0e30f, 1.0e30f);
      BeamRot r;
      const BeamSample b = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p.x, p.y}, r);
      float pr[9];
      beam_products(b, r, pr);
      run = exact_round<1024>(pr, stage, (int)threadIdx.x, run, min(1024, n - base));
    }
    float* const redf = &red[0][0][0];
    if (threadIdx.x < 9) redf[threadIdx.x] = run;
    __syncthreads();
    acc.d01 = f2{redf[0], redf[1]}; acc.d2 = redf[2];
    acc.hd = f2{redf[3], redf[4]}; acc.h22 = redf[5];
    acc.h01 = redf[6]; acc.hr = f2{redf[7], redf[8]};
  } else {
    for (int i = threadIdx.x; i < n; i += 1024) {
      const float2 p = pts[i];
      beam_accumulate<LAYOUT>(R, ex, ey, sinRot, cosRot, p.x, p.y, acc);
    }
    team_allreduce9<16>(acc, red, 0, wit, lane);
  }
  if (threadIdx.x == 0) {
    out12[0] = acc.hd.x; out12[1] = acc.h01; out12[2] = acc.hr.x;
    out12[3] = acc.h01; out12[4] = acc.hd.y; out12[5] = acc.hr.y;
    out12[6] = acc.hr.x; out12[7] = acc.hr.y; out12[8] = acc.h22;
    out12[9] = acc.d01.x; out12[10] = acc.d01.y; out12[11] = acc.d2;
  }
}

// per-beam M, dM/dx, dM/dy, rotDeriv
template <int LAYOUT>
__global__ void gn_beam_terms_kernel(const LevelView L, const float2* __restrict__ pts, int n, float ex,
                                     float ey, float eth, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sinRot, cosRot;
  sincos_f32(eth, sinRot, cosRot);
  Acc9 acc;
  acc.zero();
  BeamTerms t;
  const float2 p = pts[i];
  const LevelRegs R = level_regs<LAYOUT>(L);
  float rd = beam_accumulate<LAYOUT>(R, ex, ey, sinRot, cosRot, p.x, p.y, acc, &t);
  // out-of-map beams: the matcher's zero-texel trick yields dM = -0 where the reference returns
  // literal +0 (same sums, different sign bit); report the reference's literal values here
  const float cx = ex + (cosRot * p.x - sinRot * p.y), cy = ey + (sinRot * p.x + cosRot * p.y);
  float gx = -t.G.x, gy = -t.G.y;
  if ((cx < 0.0f) | (cx > R.limx) | (cy < 0.0f) | (cy > R.limy)) {
    t.M = gx = gy = 0.0f;
    rd = ((-sinRot * p.x - cosRot * p.y) * gx + (cosRot * p.x - sinRot * p.y) * gy);
  }
  out[i] = make_float4(t.M, gx, gy, rd);
}

// ---- f3: pose likelihood for a batch of map-frame states against ONE scan --------
// OccGridMapUtil::getLikelihoodForState (OccGridMapUtil.h:184-214): residual = sum_i (1 - M_i),
// likelihood = 1 - residual / size, with M from interpMapValue (:233-285) -- the same bilinear sample
// as the matcher without the gradient.  One wavefront per state, beams strided over the lanes, the
// texel gather and the zero-texel treatment of out-of-map beams shared with the matcher (an
// out-of-map beam reads M = 0, i.e. funval = 1, exactly the reference's `return 0.0f`).
// One residual chain in the reference's order (getResidualForState: residual += funval, i = 0 .. n-1), the
// HSM_PARITY_EXACT counterpart of the wave all-reduce below: a round of 64 beams writes its funvals to the wavefront's
// LDS row and lane 0 adds them left to right (same idea as exact_round, one term instead of nine).  Every lane returns
// the running sum that is only meaningful in lane 0.
__device__ __forceinline__ float exact_residual_round(float funval, float* __restrict__ row, int lane, float run) {
  row[lane] = funval;
  if (lane == 0) {
#pragma unroll 4
    for (int q = 0; q < 64; q += 4) {
      const float4 v = *reinterpret_cast<const float4*>(row + q);
      run += v.x;
      run += v.y;
      run += v.z;
      run += v.w;
    }
  }
  return run;
}

template <int LAYOUT, bool EXACT = false>
__global__ void __launch_bounds__(256) likelihood_kernel(const LevelView L, const float* __restrict__ states,
                                                         int batch, const float2* __restrict__ pts, int n,
                                                         float pt_scale, float* __restrict__ out_lh,
                                                         float* __restrict__ out_residual) {
  __shared__ float rows[EXACT ? 4 : 1][EXACT ? 64 : 1];
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= batch) return;
  const float ex = states[3 * b], ey = states[3 * b + 1];
  float sinRot, cosRot;
  sincos_f32(states[3 * b + 2], sinRot, cosRot);
  const LevelRegs R = level_regs<LAYOUT>(L);
  const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
  float residual = 0.0f;
  if (EXACT) {
    float* row = rows[(threadIdx.x >> 6) & 3];
    for (int base = 0; base < n; base += 64) {  // wave-uniform trip count
      const int i = base + lane;
      // padding beyond the scan: funval = +0 (M is only ever 0 .. 1, so no sum is -0 and +0 changes nothing)
      float funval = 0.0f;
      if (i < n) {
        const float2 p = pts[i];
        BeamRot r;
        const BeamSample s = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p.x * pt_scale, p.y * pt_scale}, r);
        const float M = ((s.lo.x * s.X.x + s.lo.y * s.X.y) * (s.Y.x)) + ((s.hi.x * s.X.x + s.hi.y * s.X.y) * (s.Y.y));
        funval = 1.0f - M;
      }
      residual = exact_residual_round(funval, row, lane, residual);
    }
  } else {
    for (int i = lane; i < n; i += 64) {
      const float2 p = pts[i];
      BeamRot r;
      const BeamSample s = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p.x * pt_scale, p.y * pt_scale}, r);
      const float M = ((s.lo.x * s.X.x + s.lo.y * s.X.y) * (s.Y.x)) + ((s.hi.x * s.X.x + s.hi.y * s.X.y) * (s.Y.y));
      residual += 1.0f - M;
    }
    residual = wave_allreduce(residual);
  }
  if (lane == 0) {
    if (out_lh) out_lh[b] = 1 - (residual / (float)n);
    if (out_residual) out_residual[b] = residual;  // getResidualForState (:205-221)
  }
}

// OccGridMapUtil::getCovarianceForPose (HSL/map/OccGridMapUtil.h:106-160) + getCovMatrixWorldCoords
// (:162-188): 7 sigma points around a MAP-frame pose (+-1.5 cells, +-0.05 rad, the pose itself), their
// likelihoods weight a sample mean and a 3x3 sample covariance.  One workgroup per pose, wave w scores
// sigma point w (same sampler as above), thread 0 does the 7-term statistics in the source's order.
template <int LAYOUT, bool EXACT = false>
__global__ void __launch_bounds__(448) pose_covariance_kernel(const LevelView L, const float* __restrict__ poses,
                                                              int batch, const float2* __restrict__ pts, int n,
                                                              float pt_scale, float cell_length,
                                                              float* __restrict__ out_cov_map,
                                                              float* __restrict__ out_cov_world,
                                                              float* __restrict__ out_lh7) {
  __shared__ float lh[7];
  __shared__ float rows[EXACT ? 7 : 1][EXACT ? 64 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const float deltaTransX = 1.5f, deltaTransY = 1.5f, deltaAng = 0.05f;
  const float x = poses[3 * b], y = poses[3 * b + 1], ang = poses[3 * b + 2];
  float sp[7][3] = {{x + deltaTransX, y, ang}, {x - deltaTransX, y, ang}, {x, y + deltaTransY, ang},
                    {x, y - deltaTransY, ang}, {x, y, ang + deltaAng},    {x, y, ang - deltaAng}, {x, y, ang}};
  {
    float ex = sp[0][0], ey = sp[0][1], ea = sp[0][2];
#pragma unroll
    for (int k = 1; k < 7; ++k)
      if (w == k) ex = sp[k][0], ey = sp[k][1], ea = sp[k][2];
    float sinRot, cosRot;
    sincos_f32(ea, sinRot, cosRot);
    const LevelRegs R = level_regs<LAYOUT>(L);
    const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
    float residual = 0.0f;
    if (EXACT) {
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        float funval = 0.0f;
        if (i < n) {
          const float2 p = pts[i];
          BeamRot r;
          const BeamSample s = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p.x * pt_scale, p.y * pt_scale}, r);
          const float M = ((s.lo.x * s.X.x + s.lo.y * s.X.y) * (s.Y.x)) + ((s.hi.x * s.X.x + s.hi.y * s.X.y) * (s.Y.y));
          funval = 1.0f - M;
        }
        residual = exact_residual_round(funval, rows[w], lane, residual);
      }
    } else {
      for (int i = lane; i < n; i += 64) {
        const float2 p = pts[i];
        BeamRot r;
        const BeamSample s = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{p.x * pt_scale, p.y * pt_scale}, r);
        const float M = ((s.lo.x * s.X.x + s.lo.y * s.X.y) * (s.Y.x)) + ((s.hi.x * s.X.x + s.hi.y * s.X.y) * (s.Y.y));
        residual += 1.0f - M;
      }
      residual = wave_allreduce(residual);
    }
    if (lane == 0) lh[w] = 1 - (residual / (float)n);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  // likelihoods.sum(): fixed-size 7-vector, unrolled halves (0..2) + (3..6)
  const float sum = ((lh[0] + (lh[1] + lh[2])) + ((lh[3] + lh[4]) + (lh[5] + lh[6])));
  const float invLhNormalizer = 1 / sum;
  float mean[3] = {0.0f, 0.0f, 0.0f};
  for (int i = 0; i < 7; ++i)
    for (int r = 0; r < 3; ++r) mean[r] += sp[i][r] * lh[i];
  for (int r = 0; r < 3; ++r) mean[r] *= invLhNormalizer;
  float cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // column major
  for (int i = 0; i < 7; ++i) {
    const float d[3] = {sp[i][0] - mean[0], sp[i][1] - mean[1], sp[i][2] - mean[2]};
    const float wgt = lh[i] * invLhNormalizer;
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) cov[c * 3 + r] += wgt * (d[r] * d[c]);
  }
  if (out_lh7)
    for (int i = 0; i < 7; ++i) out_lh7[7 * b + i] = lh[i];
  if (out_cov_map)
    for (int i = 0; i < 9; ++i) out_cov_map[9 * b + i] = cov[i];
  if (out_cov_world) {
    const float scaleTrans = cell_length, scaleTransSq = scaleTrans * scaleTrans;
    float* W = out_cov_world + 9 * b;  // (r,c) at c*3+r
    W[0] = cov[0] * scaleTransSq;
    W[4] = cov[4] * scaleTransSq;
    W[1] = cov[1] * scaleTransSq;  // (1,0)
    W[3] = W[1];
    W[2] = cov[2] * scaleTrans;  // (2,0)
    W[6] = W[2];
    W[5] = cov[5] * scaleTrans;  // (2,1)
    W[7] = W[5];
    W[8] = cov[8];
  }
}

}  // namespace hsm
