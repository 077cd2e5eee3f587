// hector_mi355.hip -- host runtime + C ABI (include/hector_mi355/capi.h) of the
// MI355X-native hector_mapping scan matcher.  Kernels: gn_match.h, map_update.h; the batch / team matcher forms are
// launched from match_exact_cached.hip and match_teams.hip (hsm_ctx.h says which unit holds what).
//
// The context mirrors hectorslam::MapRepMultiMap (HSL/slam_main/MapRepMultiMap.h): a
// pyramid of levels, each with its grid (log-odds + update stamps), its world<->map
// transforms (HSL/map/GridMapBase.h:265-280) and the update counters of
// OccGridMapBase (HSL/map/OccGridMapBase.h:264-266).  All planes live in HBM; the
// host keeps only scalars.  There is no CPU compute path.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
// RCCL: declarations only -- librccl is dlopen'ed by the group entry points (rccl_api), never linked.  A ROCm install without
// the RCCL development headers still builds the library: the handful of prototypes the group gather uses are then declared
// here (the stable NCCL 2.x C API; values as in nccl.h).
#if __has_include(<rccl/rccl.h>) && !defined(HSM_NO_RCCL_HEADER)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGetVersion(int* version);
}
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gn_match.h"
#include "gn_match_spec.h"
#include "hector_mi355/capi.h"
#include "hsm_ctx.h"
#include "hsm_host.h"
#include "map_update.h"

namespace hsm {
// device sin/cos sweep for the parity tests
__global__ void sincos_debug_kernel(const float* __restrict__ x, int n, float* __restrict__ s,
                                    float* __restrict__ c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sincos_f32(x[i], s[i], c[i]);
}

}  // namespace hsm

namespace {

using namespace hsm;
using namespace hsm_host;

thread_local std::string g_last_error;

}  // namespace

// every object of the library reports through this per-thread text (hsm_host.h)
int hsm_host::fail(int code, const char* what, hipError_t e) {
  char buf[512];
  if (e != hipSuccess)
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(buf, sizeof buf, "%s", what);
  g_last_error = buf;
  return code;
}

namespace {

#define HIP_TRY(expr)                                         \
  do {                                                        \
    hipError_t e__ = (expr);                                  \
    if (e__ != hipSuccess) return fail(HSM_ERR_HIP, #expr, e__); \
  } while (0)

}  // namespace


namespace {

float prob_to_log_odds(float prob) {  // GridMapLogOdds.h:196-200 (float log overload)
  float odds = prob / (1.0f - prob);
  return logf(odds);
}

// GridMapBase::setMapTransformation (GridMapBase.h:265-280) with Eigen's evaluation
// order: mapTworld = Scaling(s,s) * Translation(off); worldTmap = mapTworld.inverse()
void set_map_transformation(Level& L, float offx, float offy, float cell_length) {
  L.cell_length = cell_length;
  L.scale_to_map = 1.0f / cell_length;
  const float s = L.scale_to_map;
  Affine2 m;
  m.l00 = s;
  m.l10 = 0.0f;
  m.l01 = 0.0f;
  m.l11 = s;
  m.t0 = s * offx;
  m.t1 = s * offy;
  L.mapTworld = m;
  const float det = m.l00 * m.l11 - m.l10 * m.l01;
  const float invdet = 1.0f / det;
  Affine2 w;
  w.l00 = m.l11 * invdet;
  w.l10 = -m.l10 * invdet;
  w.l01 = -m.l01 * invdet;
  w.l11 = m.l00 * invdet;
  w.t0 = (-w.l00) * m.t0 + (-w.l01) * m.t1;
  w.t1 = (-w.l10) * m.t0 + (-w.l11) * m.t1;
  L.worldTmap = w;
}

inline void affine_apply_host(const Affine2& a, float x, float y, float& ox, float& oy) {
  ox = a.t0 + (a.l00 * x + a.l01 * y);
  oy = a.t1 + (a.l10 * x + a.l11 * y);
}

LevelRW level_rw(const Level& L) {
  LevelRW v;
  v.logodds = L.d_logodds;
  v.update_index = L.d_update_index;
  v.prob = L.d_prob;
  v.quad = L.d_quad;
  v.key_free = L.d_key_free;
  v.key_occ = L.d_key_occ;
  v.occ_bits = L.d_occ_bits;
  v.free_bytes = L.d_free_bytes;
  v.sx = L.sx;
  v.sy = L.sy;
  v.tiles_x = L.tiles_x();
  v.kf_tiles_x = key_free_tiles_x(L.sx);
  v.quad_texels = L.quad_texels();
  return v;
}

LevelView level_view(const Level& L, float pt_scale, int gn_steps) {
  LevelView v;
  v.quad = L.d_quad;
  v.prob = L.d_prob;
  v.sx = L.sx;
  v.sy = L.sy;
  v.tiles_x = L.tiles_x();
  v.quad_texels = L.quad_texels();
  v.limx = L.limx;
  v.limy = L.limy;
  v.mapTworld = L.mapTworld;
  v.worldTmap = L.worldTmap;
  v.pt_scale = pt_scale;
  v.gn_steps = gn_steps;
  return v;
}

int grid_for(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;  // grid-stride the rest (guide, Guideline 11)
  if (g < 1) g = 1;
  return (int)g;
}

int fill_level(hsm_ctx* h, Level& L) {  // GridMapBase::clear + LogOddsCell::resetGridCell
  hipLaunchKernelGGL(fill_level_kernel, dim3(grid_for(L.cells())), dim3(256), 0, h->stream, level_rw(L),
                     0.0f, -1);
  HIP_TRY(hipGetLastError());
  return HSM_OK;
}

int rebuild_probability(hsm_ctx* h, Level& L) {
  hipLaunchKernelGGL(rebuild_prob_kernel, dim3(grid_for(L.cells())), dim3(256), 0, h->stream, level_rw(L));
  if (L.d_quad)
    hipLaunchKernelGGL(rebuild_quad_kernel, dim3(grid_for(L.cells())), dim3(256), 0, h->stream, level_rw(L));
  HIP_TRY(hipGetLastError());
  return HSM_OK;
}

// Teardown never stops at a failing call (everything else still has to be released), but it must not swallow one either: HIP
// keeps the last failure per thread, and the next hipGetLastError() of an unrelated call -- the launch check of the next
// hsm_create on this thread -- would report it as its own.  The first failing call is kept for hsm_last_error(), the runtime's
// per-thread state is cleared at the end (hsm_destroy).
struct TeardownLog {
  std::string first;
  void note(const char* what, hipError_t e) {
    if (e == hipSuccess || !first.empty()) return;
    first = std::string(what) + ": " + hipGetErrorString(e);
  }
};
#define TEARDOWN(log, expr) \
  do {                      \
    hipError_t e__ = (expr); \
    if (log) (log)->note(#expr, e__); \
  } while (0)

void free_level(Level& L, TeardownLog* log = nullptr) {
  TEARDOWN(log, hipFree(L.d_logodds));
  TEARDOWN(log, hipFree(L.d_update_index));
  TEARDOWN(log, hipFree(L.d_prob));
  TEARDOWN(log, hipFree(L.d_quad));
  TEARDOWN(log, hipFree(L.d_key_free));
  TEARDOWN(log, hipFree(L.d_key_occ));
  TEARDOWN(log, hipFree(L.d_occ_bits));
  TEARDOWN(log, hipFree(L.d_free_bytes));
  L = Level();
}

int ensure_scan_capacity(float2*& buf, size_t& cap, size_t n) {
  if (n <= cap) return HSM_OK;
  if (buf) HIP_TRY(hipFree(buf));
  buf = nullptr;
  cap = 0;
  size_t want = n < 4096 ? 4096 : n + n / 2;
  HIP_TRY(hipMalloc((void**)&buf, want * sizeof(float2)));
  cap = want;
  return HSM_OK;
}

// waves per scan: enough wavefronts to fill 256 CUs x 4 SIMDs x several waves, but never
// more lanes than beams
int choose_wps(const hsm_ctx* h, int batch, int max_n) {
  if (h->wps_override > 0) return h->wps_override;
  int wps = 1;
  const long target_waves = (long)h->compute_units * 4 * 4;  // 4 waves per SIMD on every CU of THIS device (a partitioned device has fewer)
  while (wps < 16 && (long)batch * wps < target_waves && 64 * wps < max_n) wps *= 2;
  // ... but keep about five beams per lane: every extra wavefront adds LDS staging + a barrier to each
  // of the 14 dependent GN steps, which costs more than the beam loop saves (single 1081-beam scan on
  // MI355X: 72 / 58 / 53 / 61 / 79 us for 1 / 2 / 4 / 8 / 16 waves, profiles/r01/README.md)
  int lat = 1;
  while (lat < 16 && 64 * 5 * lat < max_n) lat *= 2;
  return wps < lat ? wps : lat;
}



// HSM_PARITY_AUTO (the default): EVERY match -- batched, single scan, dense scan, and the likelihood / covariance / Hessian
// entry points -- takes the reference's summation order: bit-identical to the reference CPU matcher on every entry point.
// History: round 3 chose exact order only on maps of more than 2^23 cells (a rule fitted to BASELINE's own scenes); round 4's scene
// sweep (tools/parity_scene_sweep.py, profiles/r04/parity_scene_sweep.jsonl) found the fast tree beyond 1e-4 m of the reference in
// every scene family for some set-up -- wherever the reference's own Gauss-Newton iteration has not settled -- and no property of
// the map or the batch known at launch separates those scans, so every batch went exact; single scans kept the tree on the evidence
// of 256 scans per family, of which the corridor family already failed (0.83 within 1e-4 m).  Round 5: the same argument holds for
// one scan as for 4096, so the default does not try there either.  HSM_PARITY_FAST / _RELAXED stay opt-in for callers who trade
// the guarantee for speed (profiles/r05/README.md has the prices: batch +34 % / +45 %, single 1081-beam scan ~35 vs ~95 us).
// AUTO and HSM_PARITY_EXACT launch the same kernels today (launch_match treats them alike); the distinction kept in the API is one of
// contract: AUTO promises the reference's BITS by whatever form delivers them, EXACT names the reference's order of additions.
bool auto_wants_exact(const hsm_ctx* h, const MatchParams&) { return h->auto_parity; }
// the effective summation order of the entry points that do not go through launch_match (staging decisions, likelihood,
// covariance, Hessian probes)
bool wants_exact(const hsm_ctx* h) { return h->exact || h->auto_parity; }


int launch_match_mode(hsm_ctx* h, const MatchParams& P, int max_n, hipStream_t stream, bool exact);

// HSM_PARITY_AUTO: which launches take the reference's summation order (see auto_wants_exact).  The effective mode is an
// ARGUMENT of the launch helpers -- the context's flags are never changed by a launch (hsm_parity() reads them without the
// mutex) -- and is recorded for hsm_last_launch_parity().
int launch_match(hsm_ctx* h, const MatchParams& P, int max_n, hipStream_t stream) {
  const bool exact = h->exact || auto_wants_exact(h, P);
  h->last_sorted = false;  // (the forms that take a permuted batch set it: ensure_batch_perm)
  const int rc = launch_match_mode(h, P, max_n, stream, exact);
  h->last_parity = exact ? HSM_PARITY_EXACT : (h->relaxed ? HSM_PARITY_RELAXED : HSM_PARITY_FAST);
  return rc;
}

// reference order, launches that cannot fill the chip with one wavefront per scan (single scans, small batches): one wavefront
// adds, fifteen produce one round ahead of it (gn_match_exact_dense_kernel, gn_match.h) -- a 16 k-beam match of configs[4] in
// 0.9 instead of 1.2 ms, the nine chains' own 16 384 x 14 x 8.5 cycles being 0.8
int launch_match_exact_dense(hsm_ctx* h, const MatchParams& P0, int max_n, hipStream_t stream) {
  MatchParams P = P0;
  // Round 6, opt-in (HSM_EXACT_SPEC=1): the speculative-carry form (gn_match_spec.h) -- the same sums bit for bit, the chains cut
  // into segments that run in parallel -- whenever the host knows a true bound of the scan lengths (its product scratch is sized
  // from it).  The shift rule accepts ~97 % of the segments of real chains, but on ONE CU the form is bound by what it moves
  // (16 k texel lines + 1.2 MB of products per GN step through one L1) and by a lone workgroup's ~2 us per dependent load: 2.8 ms
  // per 16 k-beam match against 0.9 for the literal chain below (profiles/r06/README.md).
  if (h->exact_spec && P.n_bound > 0 && max_n <= P.n_bound) {
    const size_t stride = spec_scratch_float4s_bound(P.n_bound);  // enough for every n <= n_bound
    const size_t need = stride * (size_t)P.batch;
    if (need > h->spec_scratch_cap) {
      HIP_TRY(hipStreamSynchronize(stream));  // (a launch in flight may still read the old block)
      if (h->d_spec_scratch) HIP_TRY(hipFree(h->d_spec_scratch));
      h->d_spec_scratch = nullptr;
      h->spec_scratch_cap = 0;
      HIP_TRY(hipMalloc((void**)&h->d_spec_scratch, need * sizeof(float4)));
      h->spec_scratch_cap = need;
    }
    P.spec_scratch = h->d_spec_scratch;
    P.spec_stride = (unsigned)stride;
    P.spec_stats = h->d_spec_stats;
    if (h->layout == kLayoutPlane)
      hipLaunchKernelGGL((gn_match_spec_kernel<kLayoutPlane>), dim3(P.batch), dim3(1024), 0, stream, P);
    else
      hipLaunchKernelGGL((gn_match_spec_kernel<kLayoutQuad>), dim3(P.batch), dim3(1024), 0, stream, P);
    HIP_TRY(hipGetLastError());
    h->last_cfg[0] = h->layout;
    h->last_cfg[1] = 16;
    h->last_cfg[2] = 1024;
    h->last_cfg[3] = P.batch;
    h->last_cfg[4] = 0;
    h->last_cfg[5] = 0;
    h->last_kernel = "gn_match_spec_kernel";
    return HSM_OK;
  }
  if (h->layout == kLayoutPlane)
    hipLaunchKernelGGL((gn_match_exact_dense_kernel<kLayoutPlane>), dim3(P.batch), dim3(1024), 0, stream, P);
  else
    hipLaunchKernelGGL((gn_match_exact_dense_kernel<kLayoutQuad>), dim3(P.batch), dim3(1024), 0, stream, P);
  HIP_TRY(hipGetLastError());
  h->last_cfg[0] = h->layout;
  h->last_cfg[1] = 16;
  h->last_cfg[2] = 1024;
  h->last_cfg[3] = P.batch;
  h->last_cfg[4] = 0;
  h->last_cfg[5] = 0;
  h->last_kernel = "gn_match_exact_dense_kernel";
  return HSM_OK;
}

int launch_match_mode(hsm_ctx* h, const MatchParams& P, int max_n, hipStream_t stream, bool exact) {
  int wps = choose_wps(h, P.batch, max_n);
  // reference order, batches of scans of up to 17 beams per lane: ALWAYS one wavefront per scan with the texel cache
  // (launch_match_exact).  Teams of wavefronts per scan -- what choose_wps picks below 4096 scans to fill the chip -- only
  // produce faster, and production is not what bounds this form: the nine chains are.  Measured (tools/batch_size_sweep.py,
  // level-0 batch of 1081-beam scans, us per launch, teams -> one wavefront per scan + chain wavefront): 16 scans 46.7 -> 36.3,
  // 1024: 80.3 -> 37.1, 2048: 92.4 -> 39.3, 3072: 134.5 -> 48.6, 3584: 135.9 -> 60.7 (without the chain wavefront).
  // (hsm_match's single scans stay on the team form: it stops its chain at the scan's last beam and keeps the endpoints in
  // registers -- 1081 beams 94 vs 92 us per call, 720 beams 74 vs 90, 360 beams 51 vs 59 through the chain-wavefront form)
  // Longer scans (rows beyond the seventeenth stream from memory in every step, one dependent round trip per row): still one
  // wavefront per scan once the batch has more scans than the device has CUs -- 2162-beam scans, 1024 / 3072 per launch: 83 / 117 us
  // against 154 / 258 for the teams; 3243 beams: 140 / 195 against 228 / 381; up to 256 scans the 16-wavefront teams are as fast
  // or faster (110-121 against 122) -- and dense scans (>= exact_dense_min beams) keep their one-workgroup-per-scan form below.
  if (exact && wps > 1 && h->wps_override == 0 && P.begin_world && !P.trace && h->layout == kLayoutQuad && h->bpl_override != 0 &&
      h->exact_cached && h->exact_chain_wave &&
      (max_n <= 17 * 64 || (P.batch > h->compute_units && !(h->exact_dense && max_n >= h->exact_dense_min))))
    wps = 1;
  if (exact && wps > 1 && h->wps_override == 0 && h->exact_dense && max_n >= h->exact_dense_min) return launch_match_exact_dense(h, P, max_n, stream);
  // one scan of the node's size through hsm_match, reference order: the on-chip speculative-carry form (gn_match_spec.h)
  if (exact && wps > 1 && h->wps_override == 0 && h->exact_spec1 && P.batch == 1 && !P.begin_world && P.n_bound > 0 && max_n <= P.n_bound &&
      max_n <= kSpec1MaxBeams) {
    if (h->layout == kLayoutPlane)
      hipLaunchKernelGGL((gn_match_spec1_kernel<kLayoutPlane>), dim3(1), dim3(1024), 0, stream, P);
    else
      hipLaunchKernelGGL((gn_match_spec1_kernel<kLayoutQuad>), dim3(1), dim3(1024), 0, stream, P);
    HIP_TRY(hipGetLastError());
    h->last_cfg[0] = h->layout;
    h->last_cfg[1] = 16;
    h->last_cfg[2] = 1024;
    h->last_cfg[3] = 1;
    h->last_cfg[4] = 0;
    h->last_cfg[5] = 0;
    h->last_kernel = "gn_match_spec1_kernel";
    return HSM_OK;
  }
  return launch_match_by_width(h, P, max_n, stream, exact, wps);
}

// the schedule of MapRepMultiMap::matchData (MapRepMultiMap.h:116-132): coarse levels
// maxIterations = 3, level 0 maxIterations = 5, each plus the unconditional first step
void fill_schedule(const hsm_ctx* h, MatchParams& P) {
  const int nl = (int)h->levels.size();
  for (int l = 0; l < nl; ++l) {
    // static_cast<float>(1.0 / pow(2.0, level)) -- a power of two, exact in fp32
    const float factor = (float)(1.0 / pow(2.0, (double)l));
    P.lv[l] = level_view(h->levels[l], factor, l == 0 ? 6 : 4);
  }
  P.first_level = nl - 1;
  P.last_level = 0;
}

int valid_level(const hsm_ctx* h, int level) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (level < 0 || level >= (int)h->levels.size()) return fail(HSM_ERR_INVALID, "level out of range");
  return HSM_OK;
}

// OccGridMapBase::updateByScan on one level (OccGridMapBase.h:121-168), host side, in two steps so that the GPU can
// start on the scan while the host is still busy:
//   prepare_level()  counters, the pose transform, the begin cell and everything the MARK pass needs -> batch.lv[]
//                    (every level with n > 0; the box fields are left empty)
//   -- the mark pass of all levels is launched here --
//   level_bbox()     the cell box of everything the scan can touch (what the dense APPLY / texel passes run over):
//                    computed on the host from the host copy of the endpoints while the mark pass runs, or derived
//                    from a finer level's box (see below)
// pts are LEVEL-0 endpoints on the device; pt_scale/origo bring them to this level.
struct LevelPrep {
  int level = -1;
  int slot = -1;          // index in batch.lv, -1 = nothing to launch for this level (empty scan)
  const float* h_pts = nullptr;
  int n = 0;
  bool derivable = false;  // set by level_bbox(): coarser levels of the same container may derive their box from this one
};

// The marks of an update -- mark bytes (dense scans), end-cell bitmap bits (keyed form) -- are cleared by the apply pass of the
// SAME update and carry no generation tag.  If that pass never ran over them (a HIP error between the two launches, or a box
// the host found empty) they would be applied by a later scan as spurious free / occupied updates.  The level remembers that
// a mark pass is outstanding; the next update on it then clears both mark planes first and widens the rows the key-wrap
// clear covers to the whole level (the failed update's keys carry an older generation and are ignored as such).
int scrub_marks(hsm_ctx* h, Level& L) {
  HIP_TRY(hipMemsetAsync(L.d_free_bytes, 0, mark_plane_bytes(L.sx, L.sy), h->stream));
  HIP_TRY(hipMemsetAsync(L.d_occ_bits, 0, ((L.cells() + 31) / 32 + 1) * sizeof(unsigned int), h->stream));
  L.key_rows[0] = 0;
  L.key_rows[1] = L.sy - 1;
  L.marks_pending = false;
  return HSM_OK;
}

int prepare_level(hsm_ctx* h, UpdateBatch& batch, LevelPrep& prep, int level, const float pose_world[3],
                  const float2* d_pts, const float* h_pts, int n, float pt_scale, const float origo_level[2]) {
  Level& L = h->levels[level];
  prep.level = level;
  prep.slot = -1;
  prep.h_pts = h_pts;
  prep.n = n;
  L.curr_mark_free = L.curr_update_index + 1;
  L.curr_mark_occ = L.curr_update_index + 2;
  float mx, my;
  affine_apply_host(L.mapTworld, pose_world[0], pose_world[1], mx, my);  // getMapCoordsPose
  const float mth = pose_world[2];
  // Translation2f(mapPose.xy) * Rotation2Df(mapPose.theta): host float sin/cos like the reference
  Affine2 T;
  const float sinA = sinf(mth), cosA = cosf(mth);
  T.l00 = cosA;
  T.l01 = -sinA;
  T.l10 = sinA;
  T.l11 = cosA;
  T.t0 = mx;
  T.t1 = my;
  float bx, by;
  affine_apply_host(T, origo_level[0], origo_level[1], bx, by);
  L.bbox[0] = L.bbox[1] = 0;
  L.bbox[2] = L.bbox[3] = -1;
  if (n > 0) {
    if (n > HSM_MAX_UPDATE_BEAMS) return fail(HSM_ERR_TOO_LARGE, "update_by_scan: more than HSM_MAX_UPDATE_BEAMS beams");
    if (L.marks_pending)
      if (int rc = scrub_marks(h, L)) return rc;
    if (++L.serial > kSerialMax) {
      // key generation wrapped (every 4095 updates of a level): clear the rows that carry keys -- the union of the update
      // boxes since the last clear, not the whole planes (an 8192^2 level would be a 512 MB memset in the middle of a
      // 40 Hz update stream)
      if (L.key_rows[1] >= L.key_rows[0]) {
        const size_t y0 = (size_t)L.key_rows[0], y1 = (size_t)L.key_rows[1];
        HIP_TRY(hipMemsetAsync(L.d_key_occ + y0 * L.sx, 0, (y1 - y0 + 1) * L.sx * sizeof(unsigned int), h->stream));
#if HSM_KEYFREE_TILE
        const size_t row_words = (size_t)key_free_tiles_x(L.sx) * 32u;  // one row of 8x4-cell tiles
        HIP_TRY(hipMemsetAsync(L.d_key_free + (y0 >> 2) * row_words, 0, ((y1 >> 2) - (y0 >> 2) + 1) * row_words * sizeof(unsigned int), h->stream));
#else
        HIP_TRY(hipMemsetAsync(L.d_key_free + y0 * L.sx, 0, (y1 - y0 + 1) * L.sx * sizeof(unsigned int), h->stream));
#endif
      }
      L.key_rows[0] = 0;
      L.key_rows[1] = -1;
      L.serial = 1;
    }
    UpdateParams P;
    P.lv = level_rw(L);
    P.pose = T;
    P.pts = d_pts;
    P.n = n;
    P.pt_scale = pt_scale;
    P.bx = (int)(bx + 0.5f);
    P.by = (int)(by + 0.5f);
    P.serial = L.serial;
    P.log_odds_free = L.log_odds_free;
    P.log_odds_occ = L.log_odds_occ;
    P.mark_free = L.curr_mark_free;
    P.mark_occ = L.curr_mark_occ;
    P.x0 = P.y0 = 0;
    P.x1 = P.y1 = -1;  // empty box until level_bbox(): the dense passes skip the level
    P.recs = nullptr;  // dense scans: set by launch_update_mark()
    prep.slot = batch.nlev;
    batch.lv[batch.nlev++] = P;
    L.marks_pending = true;  // until the apply pass over this level's box is queued (update_applied)
  }
  L.last_update_index++;     // setUpdated(), GridMapBase.h:343
  L.curr_update_index += 3;  // OccGridMapBase.h:167
  return HSM_OK;
}

// Box of level prep.level.  `finer` != nullptr: the level sees the SAME container as the finer level `finer`
// describes, scaled by a power of two -- then every fp32 value of this level's endpoint arithmetic is exactly the
// finer level's value times 2^-k (products and sums of exactly scaled operands), so cell = (int)(e * 2^-k + 0.5)
// with the finer cell (int)(e + 0.5) in [x0, x1] lies in [(x0 >> k) - 1, (x1 >> k) + 1]: a conservative box without
// touching the endpoints again (a superset only costs the dense passes a few rows of untouched cells).
//   The derivation needs the finer level to have SEEN every beam the coarser one accepts.  The low map edge breaks that:
//   (int) truncates towards zero, so level 0 keeps an end point e (cell units, before the + 0.5) with e > -1.5 while
//   level k keeps e * 2^-k > -1.5, i.e. e > -1.5 * 2^k -- a beam (or the begin cell) just outside the low x / y edge of
//   level 0 is dropped there and valid on the coarser levels (the high edge only gets stricter with k).  level 0's own
//   pass therefore records whether it rejected anything on the low side (prep.derivable); if so the coarser levels walk
//   the end points themselves.
void level_bbox(hsm_ctx* h, UpdateBatch& batch, LevelPrep& prep, const UpdateParams* finer, int shift) {
  prep.derivable = false;
  if (prep.slot < 0) return;
  Level& L = h->levels[prep.level];
  UpdateParams& P = batch.lv[prep.slot];
  const int bxi = P.bx, byi = P.by;
  const bool begin_in = bxi >= 0 && bxi < L.sx && byi >= 0 && byi < L.sy;
  int x0 = L.sx, y0 = L.sy, x1 = -1, y1 = -1;
  if (begin_in && finer) {
    if (finer->x1 >= finer->x0) {
      x0 = (finer->x0 >> shift) - 1;
      y0 = (finer->y0 >> shift) - 1;
      x1 = (finer->x1 >> shift) + 1;
      y1 = (finer->y1 >> shift) + 1;
      if (x0 < 0) x0 = 0;
      if (y0 < 0) y0 = 0;
      if (x1 > L.sx - 1) x1 = L.sx - 1;
      if (y1 > L.sy - 1) y1 = L.sy - 1;
    }
  } else if (begin_in) {
    // every in-map end cell (a Bresenham line stays inside the box of its end points).  Same fp32 expressions as
    // beam_line(); pure index work.  (Beams that end in the begin cell are skipped by the kernels; the begin cell is
    // in the box anyway.)
    const float fsx = (float)L.sx + 2.0f, fsy = (float)L.sy + 2.0f;
    const float* h_pts = prep.h_pts;
    const float pt_scale = P.pt_scale;
    bool low_reject = false;  // a beam this level drops at its low x / y edge (a coarser level may keep it)
    for (int i = 0; i < prep.n; ++i) {
      float ex, ey;
      affine_apply_host(P.pose, h_pts[2 * i] * pt_scale, h_pts[2 * i + 1] * pt_scale, ex, ey);
      ex += 0.5f;
      ey += 0.5f;
      low_reject |= (ex <= -1.0f) | (ey <= -1.0f);
      if (!(ex > -2.0f && ex < fsx && ey > -2.0f && ey < fsy)) continue;
      const int exi = (int)ex, eyi = (int)ey;
      if (exi < 0 || exi >= L.sx || eyi < 0 || eyi >= L.sy) continue;
      if (exi < x0) x0 = exi;
      if (exi > x1) x1 = exi;
      if (eyi < y0) y0 = eyi;
      if (eyi > y1) y1 = eyi;
    }
    prep.derivable = !low_reject;
  }
  if (x1 >= 0) {  // at least one beam ends inside the map
    L.bbox[0] = P.x0 = x0 < bxi ? x0 : bxi;
    L.bbox[1] = P.y0 = y0 < byi ? y0 : byi;
    L.bbox[2] = P.x1 = x1 > bxi ? x1 : bxi;
    L.bbox[3] = P.y1 = y1 > byi ? y1 : byi;
    if (L.key_rows[1] < L.key_rows[0]) {
      L.key_rows[0] = L.bbox[1];
      L.key_rows[1] = L.bbox[3];
    } else {
      if (L.bbox[1] < L.key_rows[0]) L.key_rows[0] = L.bbox[1];
      if (L.bbox[3] > L.key_rows[1]) L.key_rows[1] = L.bbox[3];
    }
    if (L.dirty[2] < L.dirty[0]) {
      for (int k = 0; k < 4; ++k) L.dirty[k] = L.bbox[k];
    } else {
      if (L.bbox[0] < L.dirty[0]) L.dirty[0] = L.bbox[0];
      if (L.bbox[1] < L.dirty[1]) L.dirty[1] = L.bbox[1];
      if (L.bbox[2] > L.dirty[2]) L.dirty[2] = L.bbox[2];
      if (L.bbox[3] > L.dirty[3]) L.dirty[3] = L.bbox[3];
    }
  }
}

// dense scans (>= merged_mark_max beams) take the byte-map form of the update (map_update.h), on every map width since round 4;
// env HSM_DENSE_BITS=0 keeps them on the keyed one-launch mark pass of the small scans
bool use_dense_bits(const hsm_ctx* h, const UpdateBatch& batch, int max_n) {
#if HSM_KEYFREE_TILE
  if (!h->dense_bits || max_n < h->merged_mark_max) return false;
  for (int i = 0; i < batch.nlev; ++i)
    if (batch.lv[i].lv.free_bytes == nullptr) return false;
  return true;
#else
  return false;
#endif
}

// pass 1 of map_update.h for all levels of the batch (grid.y = level): needs no box
int launch_update_mark(hsm_ctx* h, UpdateBatch& batch) {
  if (batch.nlev == 0) return HSM_OK;
  int max_n = 0;
  for (int i = 0; i < batch.nlev; ++i)
    if (batch.lv[i].n > max_n) max_n = batch.lv[i].n;
  const unsigned ny = (unsigned)batch.nlev;
  if (use_dense_bits(h, batch, max_n)) {
    // per-beam records: written by the end-cell pass, read by the line walk (one block of max_n records per level)
    if ((size_t)max_n > h->beam_recs_cap) {
      if (h->d_beam_recs) HIP_TRY(hipFree(h->d_beam_recs));  // (frees wait for the queued work that still reads the old block)
      h->d_beam_recs = nullptr;
      h->beam_recs_cap = 0;
      const size_t want = (size_t)max_n + (size_t)max_n / 2;
      HIP_TRY(hipMalloc((void**)&h->d_beam_recs, want * HSM_MAX_LEVELS * sizeof(BeamRec)));
      h->beam_recs_cap = want;
    }
    for (int i = 0; i < batch.nlev; ++i) batch.lv[i].recs = h->d_beam_recs + (size_t)i * h->beam_recs_cap;
    hipLaunchKernelGGL(update_mark_occ_dense_kernel, dim3((max_n + 255) / 256, ny), dim3(256), 0, h->stream, batch);
    // x extent a multiple of 8: workgroup b of every level then runs on XCD b % 8 (the kernel's beam -> XCD mapping)
    hipLaunchKernelGGL(update_mark_free_dense_kernel, dim3(mark_dense_blocks(max_n), ny), dim3(256), 0, h->stream, batch);
    HIP_TRY(hipGetLastError());
    return HSM_OK;
  }
#if defined(HSM_EXPERIMENTS)
  if (max_n >= h->merged_mark_max) {
    // rounds 1-2: dense scans on two launches (end cells, then the line walk with plain tag stores where no beam ends);
    // superseded by the byte-map form on every map width, kept for A/B builds
    hipLaunchKernelGGL(update_mark_occ_kernel, dim3((max_n + 255) / 256, ny), dim3(256), 0, h->stream, batch);
    hipLaunchKernelGGL(update_mark_free_kernel, dim3((max_n + 3) / 4, ny), dim3(256), 0, h->stream, batch);  // 4 beams (waves) per block
    HIP_TRY(hipGetLastError());
    return HSM_OK;
  }
#endif
  {
    // small scans (and HSM_DENSE_BITS=0): end-cell marks and line walks in ONE launch (keyed atomics, map_update.h) -- one
    // dependent launch less
    const unsigned occ_blocks = (unsigned)(max_n + 255) / 256;
    hipLaunchKernelGGL(update_mark_kernel, dim3(occ_blocks + (max_n + 3) / 4, ny), dim3(256), 0, h->stream, batch, occ_blocks);
  }
  HIP_TRY(hipGetLastError());
  return HSM_OK;
}

// passes 2 (+ 3) over the boxes level_bbox() filled in; levels with an empty box return at once
int launch_update_apply(hsm_ctx* h, const UpdateBatch& batch) {
  size_t max_box = 0;
  for (int i = 0; i < batch.nlev; ++i) {
    const UpdateParams& P = batch.lv[i];
    if (P.x1 < P.x0) continue;
    const size_t box = (size_t)(P.x1 - P.x0 + 2) * (size_t)(P.y1 - P.y0 + 2);
    if (box > max_box) max_box = box;
  }
  if (max_box == 0) return HSM_OK;
  const unsigned ny = (unsigned)batch.nlev;
  int max_n = 0;
  for (int i = 0; i < batch.nlev; ++i)
    if (batch.lv[i].n > max_n) max_n = batch.lv[i].n;
  if (use_dense_bits(h, batch, max_n)) {
    // one wavefront per block of 256 cells of the widened box (32 x 8 cells: two 16 x 8 mark tiles; map_update.h)
    const int g = grid_for(max_box / 4 + 4096);
    if (h->layout == kLayoutQuad)
      hipLaunchKernelGGL(update_apply_dense_kernel<true>, dim3(g, ny), dim3(256), 0, h->stream, batch);
    else
      hipLaunchKernelGGL(update_apply_dense_kernel<false>, dim3(g, ny), dim3(256), 0, h->stream, batch);
    HIP_TRY(hipGetLastError());
    return HSM_OK;
  }
  if (h->layout == kLayoutQuad && max_n < h->scatter_texels_max) {
    // the apply pass writes the texels itself: one dependent launch less for the launch-bound small scans (1081 beams,
    // 3 levels: complete 39 -> 34.5 us) and one dense pass less for the big ones (16 k beams on 8192^2: 0.51 -> 0.45 ms)
    hipLaunchKernelGGL(update_apply_kernel<true>, dim3(grid_for(max_box), ny), dim3(256), 0, h->stream, batch);
  } else {
    hipLaunchKernelGGL(update_apply_kernel<false>, dim3(grid_for(max_box), ny), dim3(256), 0, h->stream, batch);
    if (h->layout == kLayoutQuad)
      hipLaunchKernelGGL(update_texels_kernel, dim3(grid_for(max_box), ny), dim3(256), 0, h->stream, batch);
  }
  HIP_TRY(hipGetLastError());
  return HSM_OK;
}

// the apply pass of this level is queued behind its mark pass: its marks will be cleared.  A level whose box the host found
// EMPTY has no apply pass and needs none: the box is the hull of every in-map end cell, computed by level_bbox() from the same
// fp32 expressions the mark kernels evaluate (beam_line), and a beam whose end cell -- or the begin cell -- lies outside the map
// is dropped whole on the device as in the reference (OccGridMapBase.h:176-188); the non-empty case already relies on that
// equality (a mark outside the host's box would never be applied either).  Until round 4 such a level stayed "pending" and the
// NEXT update scrubbed both mark planes and widened the key rows to the whole level -- tens of MB of memset per level and update
// for as long as the robot stood outside a coarse level, or every scan was empty (round-4 advisor).  What remains pending is the
// case the flag exists for: a HIP error between the mark launch and the apply launch (the caller sees the error; the next
// update on the level scrubs first).
void update_applied(hsm_ctx* h, const UpdateBatch& batch, const LevelPrep& prep) {
  if (prep.slot < 0) return;
  h->levels[prep.level].marks_pending = false;
}

int select_device(const hsm_ctx* h) {
  HIP_TRY(hipSetDevice(h->device));
  return HSM_OK;
}

// every writer of the map queues behind a batch match that a caller-owned stream may still be running, and
// bumps the epoch the next such match orders itself behind
int order_after_foreign_match(hsm_ctx* h) {
  for (hsm_ctx::ForeignStream& f : h->foreign) {
    if (!f.pending) continue;
    // everything the caller has queued on that stream up to now (a superset of our matches); the wait captures
    // the event's state at this call, so one event serves all streams in turn
    if (!h->evt_foreign) HIP_TRY(hipEventCreateWithFlags(&h->evt_foreign, hipEventDisableTiming));
    if (hipEventRecord(h->evt_foreign, f.s) == hipSuccess) {
      HIP_TRY(hipStreamWaitEvent(h->stream, h->evt_foreign, 0));
    } else {  // the caller destroyed the stream meanwhile: its work has been flushed or is covered by a device sync
      (void)hipGetLastError();
      HIP_TRY(hipDeviceSynchronize());
    }
    f.pending = false;
  }
  if (h->foreign.size() > 16) h->foreign.clear();  // nothing pending any more: forget streams callers may have destroyed
  ++h->upd_epoch;
  return HSM_OK;
}

}  // namespace

extern "C" {

const char* hsm_last_error(void) { return g_last_error.c_str(); }
const char* hsm_version(void) { return "hector_mi355 0.1 (gfx950)"; }

int hsm_create(float map_resolution, int size_x, int size_y, unsigned levels, float start_x, float start_y,
               const hsm_opts* opts, hsm_ctx** out) {
  if (!out) return fail(HSM_ERR_INVALID, "hsm_create: out is null");
  *out = nullptr;
  if (levels < 1 || levels > HSM_MAX_LEVELS || size_x < 2 || size_y < 2 || !(map_resolution > 0.0f))
    return fail(HSM_ERR_INVALID, "hsm_create: bad map geometry");
  // the samplers address texels with a 32-bit byte offset (16 B per cell) and 24-bit multiplies
  if (size_x >= (1 << 24) || size_y >= (1 << 24) || ((size_t)size_x + 3) * ((size_t)size_y + 1) >= ((size_t)1 << 28))
    return fail(HSM_ERR_TOO_LARGE, "hsm_create: map larger than 2^28 cells");
  if ((size_x >> (levels - 1)) < 2 || (size_y >> (levels - 1)) < 2)
    return fail(HSM_ERR_INVALID, "hsm_create: too many levels for this map size");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(HSM_ERR_NO_DEVICE, "hsm_create: no HIP device (this library has no CPU path)", e);
  hsm_ctx* h = new hsm_ctx();
  if (opts && opts->device >= 0) {
    h->device = opts->device;
  } else {
    if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  }
  if (h->device >= ndev) {
    delete h;
    return fail(HSM_ERR_INVALID, "hsm_create: device ordinal out of range");
  }
  {  // (a partition of the device -- CPX mode -- reports its own CU count; 256 on a whole MI355X)
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->compute_units = cus;
  }
  int layout = opts ? opts->layout : HSM_LAYOUT_AUTO;
  if (layout == HSM_LAYOUT_AUTO) {
    const char* env = getenv("HSM_LAYOUT");
    layout = (env && strcmp(env, "plane") == 0) ? HSM_LAYOUT_PLANE : HSM_LAYOUT_QUAD;
  }
  h->layout = layout == HSM_LAYOUT_PLANE ? kLayoutPlane : kLayoutQuad;
  int wps = opts ? opts->waves_per_scan : 0;
  if (wps == 0) {
    const char* env = getenv("HSM_WPS");
    if (env) wps = atoi(env);
  }
  if (wps != 0 && wps != 1 && wps != 2 && wps != 4 && wps != 8 && wps != 16) {
    delete h;
    return fail(HSM_ERR_INVALID, "hsm_create: waves_per_scan must be 0,1,2,4,8,16");
  }
  h->wps_override = wps;
  if (const char* env = getenv("HSM_BPL")) h->bpl_override = atoi(env) == 0 ? 0 : -1;
  if (const char* env = getenv("HSM_COOP_MIN")) h->coop_min_beams = atoi(env);
  if (const char* env = getenv("HSM_COOP_TAGGED")) h->coop_tagged = atoi(env) != 0;
  if (const char* env = getenv("HSM_SPIN_WAIT")) h->spin_wait = atoi(env) != 0;
  if (const char* env = getenv("HSM_ASYNC_UPDATE")) h->async_update = atoi(env) != 0;
  if (const char* env = getenv("HSM_OVERLAP_UPLOAD")) h->overlap_upload = atoi(env) != 0;
  if (const char* env = getenv("HSM_TEXEL_CACHE")) h->texel_cache = atoi(env) != 0;
  if (const char* env = getenv("HSM_UPDATE_ZEROCOPY_MAX")) h->update_zero_copy_max = atoi(env);
  if (const char* env = getenv("HSM_PARITY")) {
    // only the four documented words change the mode; anything else ("Exact", "1", a typo) must not silently select the
    // fast tree: the context is refused
    const bool is_exact = strcmp(env, "exact") == 0, is_relaxed = strcmp(env, "relaxed") == 0, is_auto = strcmp(env, "auto") == 0;
    if (!is_exact && !is_relaxed && !is_auto && strcmp(env, "fast") != 0) {
      delete h;
      return fail(HSM_ERR_INVALID, "hsm_create: HSM_PARITY must be one of auto, fast, exact, relaxed");
    }
    h->exact = is_exact;
    h->relaxed = is_relaxed;
    h->auto_parity = is_auto;
  }
  if (const char* env = getenv("HSM_BATCH_ORDER")) {
    if (strcmp(env, "auto") == 0) h->batch_order = HSM_ORDER_AUTO;
    else if (strcmp(env, "given") == 0) h->batch_order = HSM_ORDER_GIVEN;
    else if (strcmp(env, "morton") == 0) h->batch_order = HSM_ORDER_MORTON;
    else {
      delete h;
      return fail(HSM_ERR_INVALID, "hsm_create: HSM_BATCH_ORDER must be one of auto, given, morton");
    }
  }
  if (const char* env = getenv("HSM_BATCH_ORDER_MIN")) h->batch_order_min = atoi(env);
  if (const char* env = getenv("HSM_BATCH_ORDER_REFRESH")) h->batch_order_refresh = atoi(env) > 0 ? atoi(env) : 1;
  if (const char* env = getenv("HSM_MERGED_MARK_MAX")) h->merged_mark_max = atoi(env);
  if (const char* env = getenv("HSM_SCATTER_TEXELS_MAX")) h->scatter_texels_max = atoi(env);
  if (const char* env = getenv("HSM_DENSE_BITS")) h->dense_bits = atoi(env) != 0;
  if (const char* env = getenv("HSM_EXACT_CACHED")) h->exact_cached = atoi(env) != 0;
  if (const char* env = getenv("HSM_EXACT_DENSE")) h->exact_dense = atoi(env) != 0;
  if (const char* env = getenv("HSM_EXACT_SPEC")) h->exact_spec = atoi(env) != 0;
  if (const char* env = getenv("HSM_EXACT_SPEC1")) h->exact_spec1 = atoi(env) != 0;
  if (const char* env = getenv("HSM_EXACT_DENSE_MIN")) h->exact_dense_min = atoi(env);
  if (const char* env = getenv("HSM_EXACT_CHAIN_WAVE")) h->exact_chain_wave = atoi(env);
  if (const char* env = getenv("HSM_EXACT_SPLIT_TAIL")) h->exact_split_tail = atoi(env);
  if (const char* env = getenv("HSM_WG_SYNC")) h->wg_sync = atoi(env) != 0;
#if defined(HSM_EXPERIMENTS)  // switches of forms that only an experiment build holds
  if (const char* env = getenv("HSM_EXACT_BATCH")) h->exact_batch_form = atoi(env);
  if (const char* env = getenv("HSM_EXACT_SHAPE")) h->exact_shape = atoi(env);
  if (const char* env = getenv("HSM_CACHED_WPS2")) h->cached_wps2 = atoi(env) != 0;
#endif
  if (const char* env = getenv("HSM_SPB_LARGE")) h->spb_large = atoi(env) == 8 ? 8 : 4;
  if (const char* env = getenv("HSM_XCD_CHUNK")) h->xcd_chunk = atoi(env) > 0 ? atoi(env) : 0;
  if (const char* env = getenv("HSM_XCD_CHUNK_EXACT")) h->xcd_chunk_exact = atoi(env) > 0 ? atoi(env) : 0;

#define CREATE_TRY(expr)                                   \
  do {                                                     \
    hipError_t e__ = (expr);                               \
    if (e__ != hipSuccess) {                               \
      int rc__ = fail(HSM_ERR_HIP, #expr, e__);            \
      hsm_destroy(h);                                      \
      return rc__;                                         \
    }                                                      \
  } while (0)

  CREATE_TRY(hipSetDevice(h->device));
  CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  CREATE_TRY(hipMalloc((void**)&h->d_small, kSmallFloats * sizeof(float)));
  CREATE_TRY(hipMalloc((void**)&h->d_partials, 2 * 64 * 12 * sizeof(float) + 64));  // [2][64] records of 3 x {p,p,p,tag}; + the grid-barrier counter
  CREATE_TRY(hipMemset(h->d_partials, 0, 2 * 64 * 12 * sizeof(float) + 64));
  CREATE_TRY(hipHostMalloc((void**)&h->h_small, kSmallFloats * sizeof(float),
                           hipHostMallocMapped | hipHostMallocCoherent));
  memset(h->h_small, 0, kSmallFloats * sizeof(float));

  // MapRepMultiMap ctor (MapRepMultiMap.h:48-72)
  int rx = size_x, ry = size_y;
  const float total_x = map_resolution * (float)size_x;
  const float mid_offset_x = total_x * start_x;
  const float total_y = map_resolution * (float)size_y;
  const float mid_offset_y = total_y * start_y;
  h->levels.resize(levels);
  for (unsigned i = 0; i < levels; ++i) {
    Level& L = h->levels[i];
    L.sx = rx;
    L.sy = ry;
    L.limx = (float)rx - 2.0f;  // MapDimensionProperties::setMapCellDims (:70-74)
    L.limy = (float)ry - 2.0f;
    set_map_transformation(L, mid_offset_x, mid_offset_y, map_resolution);
    L.log_odds_free = prob_to_log_odds(0.4f);  // GridMapLogOdds.h:117-118
    L.log_odds_occ = prob_to_log_odds(0.6f);
    const size_t n = L.cells();
    CREATE_TRY(hipMalloc((void**)&L.d_logodds, n * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&L.d_update_index, n * sizeof(int)));
    // the samplers point out-of-map beams at an all-zero footprint stored BEHIND the planes
    // (gn_match.h sample_fetch): one extra texel, resp. sizeX + 2 extra cells, zeroed once here
    CREATE_TRY(hipMalloc((void**)&L.d_prob, (n + (size_t)rx + 2) * sizeof(float)));
    CREATE_TRY(hipMemsetAsync(L.d_prob + n, 0, ((size_t)rx + 2) * sizeof(float), h->stream));
    if (h->layout == kLayoutQuad) {
      // HSM_LAYOUT_PLANE samples the probability plane directly (4 gathers per beam) and keeps NO texel plane:
      // 16 B/cell less memory and one dense pass less per update -- the better trade for update-heavy use
      // (single dense scans); the quad layout is the better one for batched matching
      CREATE_TRY(hipMalloc((void**)&L.d_quad, ((size_t)L.quad_texels() + 1) * sizeof(float4)));
      CREATE_TRY(hipMemsetAsync(L.d_quad + L.quad_texels(), 0, sizeof(float4), h->stream));
    }
    CREATE_TRY(hipMalloc((void**)&L.d_key_free, key_free_cells(L.sx, L.sy) * sizeof(unsigned int)));
    CREATE_TRY(hipMalloc((void**)&L.d_key_occ, n * sizeof(unsigned int)));
    CREATE_TRY(hipMemsetAsync(L.d_key_free, 0, key_free_cells(L.sx, L.sy) * sizeof(unsigned int), h->stream));
    CREATE_TRY(hipMemsetAsync(L.d_key_occ, 0, n * sizeof(unsigned int), h->stream));
    CREATE_TRY(hipMalloc((void**)&L.d_occ_bits, ((n + 31) / 32 + 1) * sizeof(unsigned int)));
    CREATE_TRY(hipMemsetAsync(L.d_occ_bits, 0, ((n + 31) / 32 + 1) * sizeof(unsigned int), h->stream));
    CREATE_TRY(hipMalloc((void**)&L.d_free_bytes, mark_plane_bytes(L.sx, L.sy)));
    CREATE_TRY(hipMemsetAsync(L.d_free_bytes, 0, mark_plane_bytes(L.sx, L.sy), h->stream));
    if (fill_level(h, L) != HSM_OK) {
      hsm_destroy(h);
      return HSM_ERR_HIP;
    }
    rx /= 2;
    ry /= 2;
    map_resolution *= 2.0f;
  }
  CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
  *out = h;
  return HSM_OK;
}

void hsm_destroy(hsm_ctx* h) {
  if (!h) return;
  TeardownLog log_, *log = &log_;
  TEARDOWN(log, hipSetDevice(h->device));
  if (h->stream) TEARDOWN(log, hipStreamSynchronize(h->stream));
  if (h->copy_stream) TEARDOWN(log, hipStreamSynchronize(h->copy_stream));
  for (Level& L : h->levels) free_level(L, log);
  TEARDOWN(log, hipFree(h->d_scan));
  TEARDOWN(log, hipFree(h->d_spec_scratch));
  for (hsm_ctx::PermBuf& b : h->perm_bufs) TEARDOWN(log, hipFree(b.d));
  TEARDOWN(log, hipFree(h->d_spec_stats));
  TEARDOWN(log, hipFree(h->d_beam_recs));
  TEARDOWN(log, hipFree(h->d_retained));
  TEARDOWN(log, hipFree(h->d_retained_alt));
  if (h->copy_evt) TEARDOWN(log, hipEventDestroy(h->copy_evt));
  if (h->copy_stream) TEARDOWN(log, hipStreamDestroy(h->copy_stream));
  if (h->h_copy_pinned) TEARDOWN(log, hipHostFree(h->h_copy_pinned));
  TEARDOWN(log, hipFree(h->d_small));
  TEARDOWN(log, hipFree(h->d_batch));
  if (h->h_hyp_pinned) TEARDOWN(log, hipHostFree(h->h_hyp_pinned));
  TEARDOWN(log, hipFree(h->d_cells));
  TEARDOWN(log, hipFree(h->d_partials));
  TEARDOWN(log, hipFree(h->d_ranges));
  TEARDOWN(log, hipFree(h->d_trig));
  TEARDOWN(log, hipFree(h->d_ingest));
  TEARDOWN(log, hipFree(h->d_occ));
  if (h->h_scan_pinned) TEARDOWN(log, hipHostFree(h->h_scan_pinned));
  if (h->evt_updates) TEARDOWN(log, hipEventDestroy(h->evt_updates));
  if (h->evt_foreign) TEARDOWN(log, hipEventDestroy(h->evt_foreign));
  if (h->h_small) TEARDOWN(log, hipHostFree(h->h_small));
  for (int k = 0; k < 2; ++k) {
    if (h->h_upd_pinned[k]) TEARDOWN(log, hipHostFree(h->h_upd_pinned[k]));
    if (h->upd_evt[k]) TEARDOWN(log, hipEventDestroy(h->upd_evt[k]));
  }
  if (h->stream) TEARDOWN(log, hipStreamDestroy(h->stream));
  delete h;
  if (!log_.first.empty()) {
    g_last_error = "hsm_destroy: " + log_.first;
    (void)hipGetLastError();  // consumed here, not by the next caller's launch check
  }
}

int hsm_reset(hsm_ctx* h) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = order_after_foreign_match(h)) return rc;
  for (Level& L : h->levels) {
    if (int rc = fill_level(h, L)) return rc;
    L.dirty[0] = L.dirty[1] = 0;  // every cell changed: the whole level is dirty for host mirrors
    L.dirty[2] = L.sx - 1;
    L.dirty[3] = L.sy - 1;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

int hsm_levels(const hsm_ctx* h) { return h ? (int)h->levels.size() : 0; }
float hsm_scale_to_map(const hsm_ctx* h) { return h ? h->levels[0].scale_to_map : 0.0f; }

int hsm_set_update_factor_free(hsm_ctx* h, float f) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  for (Level& L : h->levels) L.log_odds_free = prob_to_log_odds(f);
  return HSM_OK;
}
int hsm_set_update_factor_occupied(hsm_ctx* h, float f) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  for (Level& L : h->levels) L.log_odds_occ = prob_to_log_odds(f);
  return HSM_OK;
}
int hsm_on_map_updated(hsm_ctx* h) { return h ? HSM_OK : fail(HSM_ERR_INVALID, "null context"); }

int hsm_set_parity(hsm_ctx* h, int mode) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (mode != HSM_PARITY_FAST && mode != HSM_PARITY_EXACT && mode != HSM_PARITY_RELAXED && mode != HSM_PARITY_AUTO)
    return fail(HSM_ERR_INVALID, "hsm_set_parity: unknown mode");
  std::lock_guard<std::mutex> lk(h->mu);
  h->exact = mode == HSM_PARITY_EXACT;
  h->relaxed = mode == HSM_PARITY_RELAXED;
  h->auto_parity = mode == HSM_PARITY_AUTO;
  return HSM_OK;
}
int hsm_parity(const hsm_ctx* h) {
  if (!h) return HSM_PARITY_FAST;
  return h->exact ? HSM_PARITY_EXACT : (h->relaxed ? HSM_PARITY_RELAXED : (h->auto_parity ? HSM_PARITY_AUTO : HSM_PARITY_FAST));
}

int hsm_last_launch_parity(const hsm_ctx* h) { return h ? h->last_parity : HSM_PARITY_FAST; }

int hsm_set_batch_order(hsm_ctx* h, int order) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (order != HSM_ORDER_GIVEN && order != HSM_ORDER_MORTON && order != HSM_ORDER_AUTO) return fail(HSM_ERR_INVALID, "hsm_set_batch_order: unknown order");
  std::lock_guard<std::mutex> lk(h->mu);
  h->batch_order = order;
  for (hsm_ctx::PermBuf& b : h->perm_bufs) b.batch = 0;
  return HSM_OK;
}
int hsm_set_batch_order_refresh(hsm_ctx* h, int launches) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (launches < 1) return fail(HSM_ERR_INVALID, "hsm_set_batch_order_refresh: at least 1");
  std::lock_guard<std::mutex> lk(h->mu);
  h->batch_order_refresh = launches;
  for (hsm_ctx::PermBuf& b : h->perm_bufs) b.batch = 0;  // (the next launch computes a fresh one)
  return HSM_OK;
}
int hsm_batch_order(const hsm_ctx* h) { return h ? h->batch_order : HSM_ORDER_AUTO; }
int hsm_last_launch_sorted(const hsm_ctx* h) { return h && h->last_sorted ? 1 : 0; }

int hsm_set_clock_probe(hsm_ctx* h, unsigned long long* d_stamps4) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  h->clock_probe = d_stamps4;
  return HSM_OK;
}

int hsm_device_info(const hsm_ctx* h, int info[4]) {
  if (!h || !info) return fail(HSM_ERR_INVALID, "null argument");
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, h->device));
  info[0] = h->device;
  info[1] = p.multiProcessorCount;
  info[2] = p.clockRate;        // kHz
  info[3] = p.memoryClockRate;  // kHz
  return HSM_OK;
}

int hsm_gn_iterations_per_match(const hsm_ctx* h) {
  return h ? 6 + 4 * ((int)h->levels.size() - 1) : 0;
}
const char* hsm_last_launch_kernel(const hsm_ctx* h) { return h ? h->last_kernel : ""; }
int hsm_last_launch_config(const hsm_ctx* h, int cfg[5]) {
  if (!h || !cfg) return fail(HSM_ERR_INVALID, "null argument");
  for (int i = 0; i < 5; ++i) cfg[i] = h->last_cfg[i];
  if (h->last_cfg[5]) cfg[4] = -cfg[4];  // texel-cache form: endpoints in LDS, not VGPRs
  return HSM_OK;
}

static int match_batch_device_nolock(hsm_ctx* h, int batch, const float* d_begin_world, const float* d_pts_xy,
                                     const int* d_scan_offsets, int shared_n, float* d_out_pose,
                                     float* d_out_cov, void* stream, int n_bound = 0, const ExchangeFused* xp = nullptr) {
  if (batch < 0 || !d_begin_world || !d_out_pose || (!d_scan_offsets && shared_n < 0))
    return fail(HSM_ERR_INVALID, "hsm_match_batch_device: bad argument");
  if (batch == 0) return HSM_OK;
  if (int rc = select_device(h)) return rc;
  MatchParams P;
  memset(&P, 0, sizeof P);
  fill_schedule(h, P);
  P.batch = batch;
  P.begin_world = d_begin_world;
  P.pts = reinterpret_cast<const float2*>(d_pts_xy);
  P.offsets = d_scan_offsets;
  P.shared_n = shared_n;
  P.out_pose = d_out_pose;
  P.out_cov = d_out_cov;
  // a true bound of the scan lengths, where the host has one: a shared scan's length, or what the caller computed from host offsets
  P.n_bound = d_scan_offsets ? n_bound : shared_n;
  if (xp) P.xp = *xp;
  h->fused_exchange_done = false;
  // workgroup -> XCD mapping (gn_match.h, xcd_block): chunks dealt to the XCDs in turn balance the data-dependent
  // per-scan time; maps whose touched region outgrows the L2s keep one contiguous eighth of the batch per XCD
  P.xcd_chunk = h->levels[0].cells() <= ((size_t)1 << 23) ? h->xcd_chunk : 0;
  P.wg_sync = h->wg_sync >= 0 ? h->wg_sync : (h->levels[0].cells() > ((size_t)1 << 23) ? 1 : 0);
  P.clock_probe = h->clock_probe;
  // per-scan length is only known on the device for CSR input; shared_n doubles as the sizing HINT there
  // (callers pass the typical beams per scan, 0 = unknown).  It only picks the kernel form: every form handles
  // scans longer than the hint (the beams beyond the register/LDS-resident ones stream from memory).
  const int hint = shared_n > 0 ? shared_n : 1081;
  hipStream_t s = (hipStream_t)stream;
  if (s == h->stream) return launch_match(h, P, hint, s);
  // A caller-owned stream is not ordered against the context's own one, on which map updates are queued
  // (hsm_update_by_scan returns before they ran): order the match behind the updates queued so far, and
  // leave a marker the next update waits for, so that it does not rewrite the map under a running match.
  hsm_ctx::ForeignStream* fs = nullptr;
  for (hsm_ctx::ForeignStream& f : h->foreign)
    if (f.s == s) fs = &f;
  if (!fs) {
    h->foreign.push_back({s, 0ull, false});
    fs = &h->foreign.back();
  }
#if !defined(HSM_EXP_NO_XSTREAM_ORDER)  // (negative control of test_queued_updates_are_ordered_against_caller_streams)
  if (fs->ordered_epoch != h->upd_epoch) {  // THIS stream has not been ordered behind the latest map writes yet
    if (!h->evt_updates) HIP_TRY(hipEventCreateWithFlags(&h->evt_updates, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->evt_updates, h->stream));
    HIP_TRY(hipStreamWaitEvent(s, h->evt_updates, 0));
    fs->ordered_epoch = h->upd_epoch;
  }
#endif
  if (int rc = launch_match(h, P, hint, s)) return rc;
  // no marker here (an event record between back-to-back launches costs 2-3 us of kernel time each): the
  // next writer of the map records one on every stream with a pending match (order_after_foreign_match)
  fs->pending = true;
  return HSM_OK;
}

int hsm_match_batch_device(hsm_ctx* h, int batch, const float* d_begin_world, const float* d_pts_xy,
                           const int* d_scan_offsets, int shared_n, float* d_out_pose, float* d_out_cov,
                           void* stream) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  return match_batch_device_nolock(h, batch, d_begin_world, d_pts_xy, d_scan_offsets, shared_n, d_out_pose,
                                   d_out_cov, stream);
}

int hsm_match_batch_device_gather(hsm_ctx* h, int batch, const float* d_begin_world, const float* d_pts_xy,
                                  const int* d_scan_offsets, int shared_n, float* d_out_pose, float* d_out_cov,
                                  hsm_exchange* x, int first_row, int lag, float* d_out_all, void* stream) {
  if (!h || !x) return fail(HSM_ERR_INVALID, "null context / exchange");
  if (batch <= 0) return fail(HSM_ERR_INVALID, "hsm_match_batch_device_gather: every rank posts a non-empty shard");
  std::lock_guard<std::mutex> lk(h->mu);
  // The exchange step of this match: carried by the matcher launch itself where the form can (the exact-order batch forms: every
  // wavefront posts its pose from the kernel's epilogue, extra workgroups at the end of the grid unpack the epoch `lag` matches
  // back -- no launch of its own, nothing between two matcher launches), else one launch of the stand-alone exchange kernel behind it.
  ExchangeFused f;
  if (int rc = hsm_host::exchange_fused_begin(x, first_row, batch, lag, d_out_all, &f)) return rc;
  if (int rc = match_batch_device_nolock(h, batch, d_begin_world, d_pts_xy, d_scan_offsets, shared_n, d_out_pose, d_out_cov, stream, 0, &f))
    return rc;
  if (h->fused_exchange_done) {
    hsm_host::exchange_fused_commit(x, f);
    return HSM_OK;
  }
  return hsm_exchange_post_wait(x, d_out_pose, first_row, batch, lag, d_out_all, stream);
}

int hsm_match_batch(hsm_ctx* h, int batch, const float* begin_world, const float* pts_xy,
                    const int* scan_offsets, int shared_n, float* out_pose, float* out_cov) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (batch < 0 || !begin_world || !out_pose) return fail(HSM_ERR_INVALID, "hsm_match_batch: bad argument");
  if (batch == 0) return HSM_OK;
  const size_t total = scan_offsets ? (size_t)scan_offsets[batch] : (size_t)(shared_n > 0 ? shared_n : 0);
  if (total > 0 && !pts_xy) return fail(HSM_ERR_INVALID, "hsm_match_batch: pts_xy is null");
  const size_t b_begin = (size_t)batch * 3 * sizeof(float);
  const size_t b_pts = total * 2 * sizeof(float);
  const size_t b_offs = scan_offsets ? ((size_t)batch + 1) * sizeof(int) : 0;
  const size_t b_pose = b_begin, b_cov = (size_t)batch * 9 * sizeof(float);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t need = al(b_begin) + al(b_pts) + al(b_offs) + al(b_pose) + al(b_cov);
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (!scan_offsets) {
    // Pose hypotheses of ONE scan (a particle filter's weighting step: BASELINE configs[2]): 12 bytes in and 12 (+36) bytes out
    // per hypothesis.  The start poses and the results live in pinned, device-mapped host memory -- every wavefront reads its
    // start pose once and writes its result once, so they cross PCIe exactly once without a copy command in front of or behind the
    // launch -- and only the scan (which every wavefront reads) is copied to the device.  4096 hypotheses of a 1081-beam scan:
    // one 8.6 KB copy + the launch, against three copies, the launch and two more copies of the general path below.
    const size_t hb = al(b_begin) + al(b_pose) + al(b_cov) + al(b_pts);
    if (hb > h->h_hyp_cap) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->h_hyp_pinned) HIP_TRY(hipHostFree(h->h_hyp_pinned));
      h->h_hyp_pinned = nullptr;
      h->h_hyp_cap = 0;
      HIP_TRY(hipHostMalloc(&h->h_hyp_pinned, hb + hb / 2, hipHostMallocMapped));
      h->h_hyp_cap = hb + hb / 2;
    }
    char* hp = (char*)h->h_hyp_pinned;
    float* hp_begin = (float*)hp;
    float* hp_pose = (float*)(hp + al(b_begin));
    float* hp_cov = (float*)(hp + al(b_begin) + al(b_pose));
    float* hp_pts = (float*)(hp + al(b_begin) + al(b_pose) + al(b_cov));
    memcpy(hp_begin, begin_world, b_begin);
    if (out_cov) memcpy(hp_cov, out_cov, b_cov);  // in/out: an empty scan leaves the caller's matrices untouched (ScanMatcher.h:68,189)
    if (b_pts) memcpy(hp_pts, pts_xy, b_pts);
    char* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void**)&dp, hp, 0));
    if (int rc = ensure_scan_capacity(h->d_scan, h->d_scan_cap, total)) return rc;
    if (b_pts) HIP_TRY(hipMemcpyAsync(h->d_scan, hp_pts, b_pts, hipMemcpyHostToDevice, h->stream));
    if (int rc = match_batch_device_nolock(h, batch, (const float*)dp, (const float*)h->d_scan, nullptr, shared_n > 0 ? shared_n : 0,
                                           (float*)(dp + al(b_begin)), out_cov ? (float*)(dp + al(b_begin) + al(b_pose)) : nullptr,
                                           h->stream))
      return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    memcpy(out_pose, hp_pose, b_pose);
    if (out_cov) memcpy(out_cov, hp_cov, b_cov);
    return HSM_OK;
  }
  if (need > h->d_batch_cap) {
    if (h->d_batch) HIP_TRY(hipFree(h->d_batch));
    h->d_batch = nullptr;
    h->d_batch_cap = 0;
    HIP_TRY(hipMalloc(&h->d_batch, need));
    h->d_batch_cap = need;
  }
  char* base = (char*)h->d_batch;
  float* d_begin = (float*)base;
  float* d_pts = (float*)(base + al(b_begin));
  int* d_offs = scan_offsets ? (int*)(base + al(b_begin) + al(b_pts)) : nullptr;
  float* d_pose = (float*)(base + al(b_begin) + al(b_pts) + al(b_offs));
  float* d_cov = (float*)(base + al(b_begin) + al(b_pts) + al(b_offs) + al(b_pose));
  HIP_TRY(hipMemcpyAsync(d_begin, begin_world, b_begin, hipMemcpyHostToDevice, h->stream));
  if (b_pts) HIP_TRY(hipMemcpyAsync(d_pts, pts_xy, b_pts, hipMemcpyHostToDevice, h->stream));
  if (d_offs) HIP_TRY(hipMemcpyAsync(d_offs, scan_offsets, b_offs, hipMemcpyHostToDevice, h->stream));
  if (out_cov) HIP_TRY(hipMemcpyAsync(d_cov, out_cov, b_cov, hipMemcpyHostToDevice, h->stream));  // in/out
  int hint = shared_n;
  if (scan_offsets) {
    hint = 0;
    for (int i = 0; i < batch; ++i) {
      const int ni = scan_offsets[i + 1] - scan_offsets[i];
      if (ni > hint) hint = ni;
    }
  }
  if (int rc = match_batch_device_nolock(h, batch, d_begin, d_pts, d_offs, hint, d_pose,
                                         out_cov ? d_cov : nullptr, h->stream, scan_offsets ? hint : 0))
    return rc;
  HIP_TRY(hipMemcpyAsync(out_pose, d_pose, b_pose, hipMemcpyDeviceToHost, h->stream));
  if (out_cov) HIP_TRY(hipMemcpyAsync(out_cov, d_cov, b_cov, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

// Largest scan the matcher keeps entirely in registers (16 waves x 64 lanes x 17 beams): such a scan is
// read exactly once per match, so the kernel can fetch it directly from pinned host memory over PCIe
// and the host entry needs no H2D copy at all.
constexpr int kMaxRegisterResidentBeams = 16 * 64 * 17;

// One scan on the levels selected in P.  `pts` is a device-accessible pointer (device memory or pinned
// mapped host memory), level-0 units.  Latency path of the ROS node: ONE kernel launch and one stream
// synchronise -- the start estimate travels in the kernel arguments and the kernel writes pose, H and
// the optional hook trace straight into the pinned h_small block.
// Completion of a single-scan match.  The kernel's last act is a system-scope release store of `seq` into
// the pinned result block, AFTER pose / cov / trace: polling that word returns the results a few
// microseconds before the end-of-kernel signal would (the queue's completion interrupt path is most of
// what hipStreamSynchronize waits for on a 25 us kernel).  Bounded: after 2 ms without the word the normal
// stream synchronisation takes over, which also surfaces a faulted kernel.  Later work on the stream
// stays ordered behind the kernel as usual.
static int wait_single_scan(hsm_ctx* h, unsigned seq) {
  if (h->spin_wait) {
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(h->h_small + kDoneFlagOff);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 1;; ++it) {
      if (*flag == seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        return HSM_OK;
      }
      __builtin_ia32_pause();
      if ((it & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

static int match_single(hsm_ctx* h, MatchParams& P, const float begin_world[3], const float2* pts, int n,
                        float out_pose_world[3], float cov[9], float* trace = nullptr, int trace_steps = 0) {
  float* hs = h->h_small;
  float* hs_dev = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&hs_dev, hs, 0));
  P.batch = 1;
  P.begin_world = nullptr;
  P.begin_inline[0] = begin_world[0];
  P.begin_inline[1] = begin_world[1];
  P.begin_inline[2] = begin_world[2];
  P.pts = pts;
  P.offsets = nullptr;
  P.shared_n = n;
  P.n_bound = n;
  P.out_pose = hs_dev + 3;
  P.out_cov = hs_dev + 6;
  P.trace = trace_steps > 0 ? hs_dev + kTraceOff : nullptr;
  const unsigned seq = ++h->done_seq;
  P.done_flag = h->spin_wait ? reinterpret_cast<unsigned*>(hs_dev + kDoneFlagOff) : nullptr;
  P.done_seq = seq;
  P.err_flag = reinterpret_cast<unsigned*>(hs_dev + kErrFlagOff);
  P.coop_mute_block = h->coop_mute_block;
  P.clock_probe = h->clock_probe;
  bool coop_now = n >= h->coop_min_beams && h->wps_override == 0 && !wants_exact(h);
  if (coop_now && h->coop_skip > 0) {  // backing off after an exchange timeout: this match goes straight to the one-workgroup form
    --h->coop_skip;
    coop_now = false;
  }
  const bool coop_tried = coop_now;
  if (coop_now) {
    // one dense scan: spread it over K workgroups of one cooperative launch (gn_match.h); the exact-order
    // form keeps the scan on one workgroup -- its nine summation chains are sequential anyway
    // one beam per lane.  16 k beams, matchData us for K = 16 / 24 / 32 / 64 workgroups: 79.8 / 71 / 66-70 / 64 with round 2's grid
    // barrier; 70 (24) / 67-71 (31) / 76 (48) / 63-64 (64) with the tagged exchange (profiles/r03/README.md)
    int K = (n + 255) / 256;
#if defined(HSM_EXPERIMENTS)
    if (const char* env = getenv("HSM_COOP_K")) K = atoi(env);
#endif
    if (K > 64) K = 64;
    if (K < 2) K = 2;
    float* partials = h->d_partials;
    unsigned* bar_counter = reinterpret_cast<unsigned*>(h->d_partials + 2 * 64 * 12);
    unsigned bar_base = h->coop_bar_base;
    void* args[] = {(void*)&P, (void*)&partials, (void*)&bar_counter, (void*)&bar_base};
    const void* fn = h->layout == kLayoutPlane ? (const void*)gn_match_coop_kernel<kLayoutPlane, false>
                                                : (const void*)gn_match_coop_kernel<kLayoutQuad, false>;
    // The tagged-record exchange (default) has no grid barrier: a workgroup that is not resident yet only delays the others'
    // polls, which are bounded (a record that never arrives turns into an error return, err_flag) -- so it is an ORDINARY
    // launch: K <= 64 workgroups of 256 lanes are co-resident on an idle 256-CU device, and on a busy one they become so as
    // soon as other kernels retire.  (Round 3 launched it through hipLaunchCooperativeKernel: that goes through the
    // device's cooperative queue -- a slower launch, and a process that has used it segfaults in the runtime's exit
    // handlers when it runs under rocprofv3, profiles/r04/README.md.)  The counter-barrier form (HSM_COOP_TAGGED=0) spins
    // without a bound and keeps the cooperative launch's co-residency guarantee.
    hipError_t le;
    if (h->coop_tagged) {
      if (h->layout == kLayoutPlane)
        hipLaunchKernelGGL((gn_match_coop_kernel<kLayoutPlane, true>), dim3(K), dim3(256), 0, h->stream, P, partials, bar_counter, bar_base);
      else
        hipLaunchKernelGGL((gn_match_coop_kernel<kLayoutQuad, true>), dim3(K), dim3(256), 0, h->stream, P, partials, bar_counter, bar_base);
      le = hipGetLastError();
    } else {
      le = hipLaunchCooperativeKernel(fn, dim3(K), dim3(256), args, 0, h->stream);
    }
    if (le == hipSuccess) {
      unsigned steps = 0;
      for (int l = P.first_level; l >= P.last_level; --l) steps += (unsigned)P.lv[l].gn_steps;
      h->coop_bar_base += (unsigned)K * steps;  // one arrival per workgroup per GN step
      h->last_cfg[0] = h->layout;
      h->last_cfg[1] = -K;  // negative: K cooperating workgroups instead of waves per scan
      h->last_cfg[2] = 256;
      h->last_cfg[3] = K;
      h->last_cfg[4] = 0;
      h->last_kernel = "gn_match_coop_kernel";
      h->last_parity = HSM_PARITY_FAST;
    } else {
      // the runtime could not guarantee co-residency (device busy with other work): the one-workgroup
      // matcher computes the same thing on one CU
      (void)hipGetLastError();
      if (int rc = launch_match(h, P, n, h->stream)) return rc;
    }
  } else if (int rc = launch_match(h, P, n, h->stream)) {
    return rc;
  }
  if (int rc = wait_single_scan(h, seq)) return rc;
  h->queued_update = false;  // the match kernel was the last thing on `stream`, and it has completed
  if (*reinterpret_cast<volatile unsigned*>(hs + kErrFlagOff) == seq) {
    // The multi-workgroup matcher's tagged exchange gave up waiting for a record: its K workgroups were not co-resident for
    // ~2^22 polls (a device shared with another process, or a long kernel of another stream holding the CUs -- an ordinary
    // launch carries no co-residency guarantee).  No pose was written.  The scan is matched again by the one-workgroup
    // matcher, which needs no other workgroup to make progress -- same sums in a different tree, so the caller gets a pose
    // within the fast mode's bar instead of an error (round-4 advisor); hsm_last_launch_config() then reports that form.
    const unsigned seq2 = ++h->done_seq;
    P.done_seq = seq2;
    if (int rc = launch_match(h, P, n, h->stream)) return rc;
    if (int rc = wait_single_scan(h, seq2)) return rc;
    ++h->coop_fallbacks;
    h->coop_backoff = h->coop_backoff ? (h->coop_backoff < 1024 ? 2 * h->coop_backoff : 1024) : 1;
    h->coop_skip = h->coop_backoff;
    {
      char b[200];
      snprintf(b, sizeof b, "hsm_match: the multi-workgroup matcher's exchange timed out (device shared or busy); re-ran on one workgroup, "
               "skipping that form for the next %u dense matches", h->coop_skip);
      g_last_error = b;  // (not an error return: the pose is valid; the text tells who asks why a match took long)
    }
    if (*reinterpret_cast<volatile unsigned*>(hs + kErrFlagOff) == seq2)
      return fail(HSM_ERR_HIP, "hsm_match: exchange timeout flagged by the one-workgroup matcher (cannot happen: it has no exchange)");
  }
  else if (coop_tried)
    h->coop_backoff = 0;  // an exchange completed: the device is ours again
  for (int i = 0; i < trace_steps * 12; ++i) trace[i] = hs[kTraceOff + i];
  out_pose_world[0] = hs[3];
  out_pose_world[1] = hs[4];
  out_pose_world[2] = hs[5];
  if (n != 0 && cov)
    for (int i = 0; i < 9; ++i) cov[i] = hs[6 + i];
  return HSM_OK;
}

// Does the matcher read the endpoints of a single n-beam scan exactly ONCE?  Then they can stay in pinned, device-mapped host
// memory (no H2D copy command in front of the kernel); otherwise -- re-read in every GN step -- they must live in device memory.
//   tree summation: the register-resident forms (BPL > 0), unless the multi-workgroup dense matcher takes the scan;
//   reference order (round 5): the team form keeps a scan of at most kExactGroupRounds rounds in registers across all levels and
//     steps (gn_match_kernel: xq_resident) -- every single scan below the dense threshold --, the producers-ahead form a scan of
//     at most two of its rounds
static bool scan_is_read_once(const hsm_ctx* h, int n) {
  if (!wants_exact(h))
    return n <= kMaxRegisterResidentBeams && h->bpl_override != 0 && (n < h->coop_min_beams || h->wps_override != 0);
  const int wps = choose_wps(h, 1, n);
  if (wps > 1 && h->wps_override == 0 && h->exact_dense && n >= h->exact_dense_min)
    return !h->exact_spec && n <= 2 * kDenseRound;  // (the speculative-carry form re-reads the endpoints in every GN step)
  return n <= kExactGroupRounds * 64 * wps;
}

// stage a host scan where the matcher can read it: pinned mapped host memory when it will be read
// once (register resident), device memory otherwise
static int stage_scan(hsm_ctx* h, const float* pts_xy, int n, float2*& d_buf, size_t& d_cap, const float2** out) {
  // (a dense scan for the multi-workgroup matcher is re-read every GN step: it must live in device memory)
  // (and so does the exact-order form)
  if (scan_is_read_once(h, n)) {
    if (!h->h_scan_pinned || (size_t)n > h->h_scan_pinned_cap) {  // (also the empty first scan of a fresh context)
      if (h->h_scan_pinned) HIP_TRY(hipHostFree(h->h_scan_pinned));
      h->h_scan_pinned = nullptr;
      h->h_scan_pinned_cap = 0;
      const size_t want = n < 4096 ? 4096 : (size_t)n + n / 2;
      HIP_TRY(hipHostMalloc((void**)&h->h_scan_pinned, want * sizeof(float2), hipHostMallocMapped));
      h->h_scan_pinned_cap = want;
    }
    if (n > 0) memcpy(h->h_scan_pinned, pts_xy, (size_t)n * sizeof(float2));
    float2* dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void**)&dev, h->h_scan_pinned, 0));
    *out = dev;
    return HSM_OK;
  }
  if (int rc = ensure_scan_capacity(d_buf, d_cap, (size_t)n)) return rc;
  if (n > 0) HIP_TRY(hipMemcpyAsync(d_buf, pts_xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  *out = d_buf;
  return HSM_OK;
}

// The retained scan of matchData when it has to live in device memory (dense scans, exact order): uploaded on the copy
// stream into the buffer the update kernels of the scan BEFORE are not reading, so that the copy runs while those kernels
// still occupy `stream` (in a match + update loop the upload of scan t + 1 used to wait behind the update of scan t: ~10 us of
// every configs[4] step).  Safe with two buffers: the kernels that read buffer A (match t, update t) are all ordered before
// match t + 1 on `stream`, and hsm_match returns only when match t + 1 has completed -- so when the upload of scan t + 2 is
// issued into A nothing reads it any more.  The staging block is pinned (a pageable hipMemcpyAsync would block the host
// until everything queued on its stream has completed) and free again for the same reason.
static int stage_scan_overlapped(hsm_ctx* h, const float* pts_xy, int n, const float2** out) {
  if (!h->copy_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->copy_evt, hipEventDisableTiming));
  }
  std::swap(h->d_retained, h->d_retained_alt);
  std::swap(h->d_retained_cap, h->d_retained_alt_cap);
  if (int rc = ensure_scan_capacity(h->d_retained, h->d_retained_cap, (size_t)n)) return rc;
  if ((size_t)n > h->h_copy_pinned_cap) {
    HIP_TRY(hipStreamSynchronize(h->copy_stream));
    if (h->h_copy_pinned) HIP_TRY(hipHostFree(h->h_copy_pinned));
    h->h_copy_pinned = nullptr;
    h->h_copy_pinned_cap = 0;
    const size_t want = (size_t)n + (size_t)n / 2;
    HIP_TRY(hipHostMalloc((void**)&h->h_copy_pinned, want * sizeof(float2), hipHostMallocDefault));
    h->h_copy_pinned_cap = want;
  }
  memcpy(h->h_copy_pinned, pts_xy, (size_t)n * sizeof(float2));
  HIP_TRY(hipMemcpyAsync(h->d_retained, h->h_copy_pinned, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, h->copy_stream));
  HIP_TRY(hipEventRecord(h->copy_evt, h->copy_stream));
  HIP_TRY(hipStreamWaitEvent(h->stream, h->copy_evt, 0));
  *out = h->d_retained;
  return HSM_OK;
}

static int match_impl(hsm_ctx* h, const float begin_world[3], const float* pts_xy, int n, const float origo[2],
                      float out_pose_world[3], float cov[9], float* trace, int trace_steps,
                      const float2* d_prestaged = nullptr);

int hsm_match(hsm_ctx* h, const float begin_world[3], const float* pts_xy, int n, const float origo[2],
              float out_pose_world[3], float cov[9]) {
  return match_impl(h, begin_world, pts_xy, n, origo, out_pose_world, cov, nullptr, 0);
}

int hsm_match_trace(hsm_ctx* h, const float begin_world[3], const float* pts_xy, int n, const float origo[2],
                    float out_pose_world[3], float cov[9], float* trace, int trace_cap_steps, int* steps_written) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  const int steps = hsm_gn_iterations_per_match(h);
  if (!trace || !steps_written || trace_cap_steps < steps)
    return fail(HSM_ERR_INVALID, "hsm_match_trace: trace buffer smaller than hsm_gn_iterations_per_match()");
  *steps_written = n > 0 ? steps : 0;
  return match_impl(h, begin_world, pts_xy, n, origo, out_pose_world, cov, trace, n > 0 ? steps : 0);
}

static int match_impl(hsm_ctx* h, const float begin_world[3], const float* pts_xy, int n, const float origo[2],
                      float out_pose_world[3], float cov[9], float* trace, int trace_steps,
                      const float2* d_prestaged) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (!begin_world || !out_pose_world || n < 0 || (n > 0 && !pts_xy))
    return fail(HSM_ERR_INVALID, "hsm_match: bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  // DataContainer::setFrom keeps scaled copies for the coarse levels (MapRepMultiMap.h:127);
  // here: one level-0 copy, scaled by 2^-level on use
  if (h->levels.size() > 1) {
    h->retained_pts.assign(pts_xy, pts_xy + 2 * (size_t)n);
    h->retained_origo[0] = origo ? origo[0] : 0.0f;
    h->retained_origo[1] = origo ? origo[1] : 0.0f;
    h->retained_valid = true;
  }
  const float2* pts = d_prestaged;
  if (!pts) {
    // (the same rule as stage_scan's: scans the matcher reads once stay in pinned host memory)
    const bool to_device = !scan_is_read_once(h, n);
    // ... and only behind an update that was queued and not waited for (the match + update loop): on an idle stream the
    // extra hop through the copy stream's event costs ~10 us of latency and hides nothing (asking the runtime with
    // hipStreamQuery costs half of what the overlap gains: 0.1855 against 0.179 ms per configs[4] step)
    if (to_device && h->overlap_upload && n > 0 && h->queued_update) {
      if (int rc = stage_scan_overlapped(h, pts_xy, n, &pts)) return rc;
    } else if (int rc = stage_scan(h, pts_xy, n, h->d_retained, h->d_retained_cap, &pts)) {
      return rc;
    }
  }
  // the device copy of the retained scan is (re)uploaded lazily by the next update when the
  // matcher read the scan from pinned host memory
  h->d_retained_current = h->levels.size() > 1 && pts == h->d_retained;
  MatchParams P;
  memset(&P, 0, sizeof P);
  fill_schedule(h, P);
  return match_single(h, P, begin_world, pts, n, out_pose_world, cov, trace, trace_steps);
}

int hsm_match_level(hsm_ctx* h, int level, const float begin_world[3], const float* pts_level_xy, int n,
                    int max_iterations, float out_pose_world[3], float cov[9]) {
  if (int rc = valid_level(h, level)) return rc;
  if (!begin_world || !out_pose_world || n < 0 || max_iterations < 0 || (n > 0 && !pts_level_xy))
    return fail(HSM_ERR_INVALID, "hsm_match_level: bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  const float2* pts = nullptr;
  if (int rc = stage_scan(h, pts_level_xy, n, h->d_scan, h->d_scan_cap, &pts)) return rc;
  MatchParams P;
  memset(&P, 0, sizeof P);
  P.lv[level] = level_view(h->levels[level], 1.0f, 1 + max_iterations);
  P.first_level = level;
  P.last_level = level;
  return match_single(h, P, begin_world, pts, n, out_pose_world, cov);
}

static int update_impl(hsm_ctx* h, const float pose_world[3], const float* pts_xy, int n, const float origo[2],
                       const float2* d_prestaged);

int hsm_update_by_scan(hsm_ctx* h, const float pose_world[3], const float* pts_xy, int n, const float origo[2]) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (!pose_world || n < 0 || (n > 0 && !pts_xy)) return fail(HSM_ERR_INVALID, "hsm_update_by_scan: bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  return update_impl(h, pose_world, pts_xy, n, origo, nullptr);
}

// pts_xy: host copy of the endpoints (always needed: the touched bounding box is computed on the host);
// d_prestaged: the same endpoints already on the device, or nullptr to upload them
static int update_impl(hsm_ctx* h, const float pose_world[3], const float* pts_xy, int n, const float origo[2],
                       const float2* d_prestaged) {
  if (int rc = select_device(h)) return rc;
  if (int rc = order_after_foreign_match(h)) return rc;
  const float zero[2] = {0.0f, 0.0f};
  const float* o = origo ? origo : zero;
  // level 0: the caller's container
  const float2* d_level0 = d_prestaged;
  int slot = -1;
  // The usual flow updates with the scan that was just matched.  When the matcher put that scan into device memory (dense
  // scans, multi-level maps: d_retained) and the caller hands over the same endpoints, they are already where the update
  // kernels read them: no second upload (131 KB from pageable memory for a 16 k-beam scan, which the host waits for).
  const int rn0 = h->retained_valid ? (int)(h->retained_pts.size() / 2) : 0;
  const bool same_as_matched = !d_level0 && h->d_retained_current && rn0 == n && n > 0 &&
                               memcmp(h->retained_pts.data(), pts_xy, (size_t)n * sizeof(float2)) == 0;
  if (same_as_matched) d_level0 = h->d_retained;
  if (!d_level0 && h->async_update) {
    // stage the endpoints in pinned memory (a pageable hipMemcpyAsync would block the host until the copy
    // -- and everything queued before it -- has completed); small scans are then read in place over PCIe
    // (each endpoint is read twice), dense ones copied to the device by a copy the host does not wait for
    slot = h->upd_slot;
    h->upd_slot ^= 1;
    if (h->upd_busy[slot]) {
      HIP_TRY(hipEventSynchronize(h->upd_evt[slot]));
      h->upd_busy[slot] = false;
    }
    if ((size_t)n > h->h_upd_cap[slot]) {
      if (h->h_upd_pinned[slot]) HIP_TRY(hipHostFree(h->h_upd_pinned[slot]));
      h->h_upd_pinned[slot] = nullptr;
      h->h_upd_cap[slot] = 0;
      const size_t want = n < 4096 ? 4096 : (size_t)n + n / 2;
      HIP_TRY(hipHostMalloc((void**)&h->h_upd_pinned[slot], want * sizeof(float2), hipHostMallocMapped));
      h->h_upd_cap[slot] = want;
    }
    if (!h->upd_evt[slot]) HIP_TRY(hipEventCreateWithFlags(&h->upd_evt[slot], hipEventDisableTiming));
    if (n > 0) memcpy(h->h_upd_pinned[slot], pts_xy, (size_t)n * sizeof(float2));
    if (n <= h->update_zero_copy_max) {
      float2* dev = nullptr;
      if (n > 0) HIP_TRY(hipHostGetDevicePointer((void**)&dev, h->h_upd_pinned[slot], 0));
      d_level0 = n > 0 ? dev : h->d_scan;
    } else {
      if (int rc = ensure_scan_capacity(h->d_scan, h->d_scan_cap, (size_t)n)) return rc;
      HIP_TRY(hipMemcpyAsync(h->d_scan, h->h_upd_pinned[slot], (size_t)n * sizeof(float2), hipMemcpyHostToDevice,
                             h->stream));
      d_level0 = h->d_scan;
    }
  } else if (!d_level0) {
    if (int rc = ensure_scan_capacity(h->d_scan, h->d_scan_cap, (size_t)n)) return rc;
    if (n > 0)
      HIP_TRY(hipMemcpyAsync(h->d_scan, pts_xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    d_level0 = h->d_scan;
  }
  UpdateBatch batch;
  batch.nlev = 0;
  LevelPrep prep[HSM_MAX_LEVELS];
  if (int rc = prepare_level(h, batch, prep[0], 0, pose_world, d_level0, pts_xy, n, 1.0f, o)) return rc;
  // coarse levels: the containers retained by the last matchData (MapRepMultiMap.h:143)
  const int rn = h->retained_valid ? (int)(h->retained_pts.size() / 2) : 0;
  bool coarse_same_container = false;  // the usual flow: update with the container that was just matched
  if (h->levels.size() > 1) {
    const float2* d_coarse = nullptr;
    if (same_as_matched) {
      d_coarse = d_level0;
      coarse_same_container = h->retained_origo[0] == o[0] && h->retained_origo[1] == o[1];
    } else if (rn == n && n > 0 && !h->d_retained_current &&
               memcmp(h->retained_pts.data(), pts_xy, (size_t)n * sizeof(float2)) == 0) {
      d_coarse = d_level0;
      coarse_same_container = h->retained_origo[0] == o[0] && h->retained_origo[1] == o[1];
    } else {
      if (rn > 0 && !h->d_retained_current) {
        if (int rc = ensure_scan_capacity(h->d_retained, h->d_retained_cap, (size_t)rn)) return rc;
        HIP_TRY(hipMemcpyAsync(h->d_retained, h->retained_pts.data(), (size_t)rn * sizeof(float2),
                               hipMemcpyHostToDevice, h->stream));
        h->d_retained_current = true;
      }
      d_coarse = h->d_retained;  // read only after a possible (re)allocation above
    }
    for (size_t l = 1; l < h->levels.size(); ++l) {
      const float factor = (float)(1.0 / pow(2.0, (double)l));
      const float ol[2] = {h->retained_origo[0] * factor, h->retained_origo[1] * factor};  // setFrom :48
      if (int rc = prepare_level(h, batch, prep[l], (int)l, pose_world, d_coarse, h->retained_pts.data(), rn, factor, ol))
        return rc;
    }
  }
  // the GPU starts marking (all levels, one launch) while the host works out the boxes of the dense passes
  if (int rc = launch_update_mark(h, batch)) return rc;
  level_bbox(h, batch, prep[0], nullptr, 0);
  for (size_t l = 1; l < h->levels.size(); ++l) {
    const bool derive = coarse_same_container && prep[0].slot >= 0 && prep[0].derivable && h->levels[l].sx == (h->levels[0].sx >> l) &&
                        h->levels[l].sy == (h->levels[0].sy >> l);
    level_bbox(h, batch, prep[l], derive ? &batch.lv[prep[0].slot] : nullptr, (int)l);
  }
  if (int rc = launch_update_apply(h, batch)) return rc;
  for (size_t l = 0; l < h->levels.size(); ++l) update_applied(h, batch, prep[l]);
  h->queued_update = h->async_update;
  if (slot >= 0) {
    HIP_TRY(hipEventRecord(h->upd_evt[slot], h->stream));
    h->upd_busy[slot] = true;
  }
  if (!h->async_update) HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

int hsm_synchronize(hsm_ctx* h) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->upd_busy[0] = h->upd_busy[1] = false;
  h->queued_update = false;
  return HSM_OK;
}

int hsm_update_by_scan_level(hsm_ctx* h, int level, const float pose_world[3], const float* pts_level_xy, int n,
                             const float origo_level[2]) {
  if (int rc = valid_level(h, level)) return rc;
  if (!pose_world || n < 0 || (n > 0 && !pts_level_xy))
    return fail(HSM_ERR_INVALID, "hsm_update_by_scan_level: bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = order_after_foreign_match(h)) return rc;
  const float zero[2] = {0.0f, 0.0f};
  if (int rc = ensure_scan_capacity(h->d_scan, h->d_scan_cap, (size_t)n)) return rc;
  if (n > 0)
    HIP_TRY(hipMemcpyAsync(h->d_scan, pts_level_xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  UpdateBatch batch;
  batch.nlev = 0;
  LevelPrep prep;
  if (int rc = prepare_level(h, batch, prep, level, pose_world, h->d_scan, pts_level_xy, n, 1.0f,
                             origo_level ? origo_level : zero))
    return rc;
  if (int rc = launch_update_mark(h, batch)) return rc;
  level_bbox(h, batch, prep, nullptr, 0);
  if (int rc = launch_update_apply(h, batch)) return rc;
  update_applied(h, batch, prep);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

// device buffers of the ingestion entries: raw input (3 floats per element covers ranges and Point32
// clouds; the int behind it is the survivor count), the sensor-geometry table (16 B per beam covers the
// float2 and the double2 variant) and the container
static int ensure_ingest_capacity(hsm_ctx* h, int n) {
  if (h->d_ranges && (size_t)n <= h->ingest_cap) return HSM_OK;
  (void)hipFree(h->d_ranges);
  (void)hipFree(h->d_trig);
  (void)hipFree(h->d_ingest);
  h->d_ranges = nullptr;
  h->d_trig = nullptr;
  h->d_ingest = nullptr;
  h->ingest_cap = 0;
  h->trig_n = -1;
  const size_t want = n < 2048 ? 2048 : (size_t)n + n / 2;
  HIP_TRY(hipMalloc((void**)&h->d_ranges, 3 * want * sizeof(float) + sizeof(int)));
  HIP_TRY(hipMalloc((void**)&h->d_trig, want * sizeof(double2)));
  HIP_TRY(hipMalloc((void**)&h->d_ingest, want * sizeof(float2)));
  h->ingest_cap = want;
  return HSM_OK;
}

static int* ingest_count_ptr(hsm_ctx* h) { return reinterpret_cast<int*>(h->d_ranges + 3 * h->ingest_cap); }

// fetch the survivor count + the container the kernel on h->stream just produced
static int finish_ingest(hsm_ctx* h, float* out_pts_xy, int* out_n) {
  int m = 0;
  HIP_TRY(hipMemcpyAsync(&m, ingest_count_ptr(h), sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->h_ingest.resize(2 * (size_t)m);
  if (m > 0) HIP_TRY(hipMemcpy(h->h_ingest.data(), h->d_ingest, (size_t)m * sizeof(float2), hipMemcpyDeviceToHost));
  h->ingest_n = m;
  if (out_pts_xy && m > 0) memcpy(out_pts_xy, h->h_ingest.data(), (size_t)m * sizeof(float2));
  if (out_n) *out_n = m;
  return HSM_OK;
}

int hsm_ingest_laser_scan(hsm_ctx* h, const float* ranges, int n, float angle_min, float angle_increment,
                          float range_min, float range_max, float scale_to_map, float* out_pts_xy, int* out_n) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (n < 0 || (n > 0 && !ranges) || n > HSM_MAX_UPDATE_BEAMS)
    return fail(HSM_ERR_INVALID, "hsm_ingest_laser_scan: bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = ensure_ingest_capacity(h, n)) return rc;
  if (h->trig_kind != 0 || h->trig_n != n || h->trig_a0 != angle_min || h->trig_inc != angle_increment) {
    // the node's running fp32 angle and its float cos/sin (HectorMappingRos.cpp:487,502,505): sensor
    // constants, evaluated once per geometry on the host exactly as the node does
    std::vector<float> t(2 * (size_t)n);
    float angle = angle_min;
    for (int i = 0; i < n; ++i) {
      t[2 * i] = cosf(angle);
      t[2 * i + 1] = sinf(angle);
      angle += angle_increment;
    }
    if (n > 0) HIP_TRY(hipMemcpy(h->d_trig, t.data(), (size_t)n * sizeof(float2), hipMemcpyHostToDevice));
    h->trig_kind = 0;
    h->trig_n = n;
    h->trig_a0 = angle_min;
    h->trig_inc = angle_increment;
  }
  if (n > 0) HIP_TRY(hipMemcpyAsync(h->d_ranges, ranges, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  const float maxRangeForContainer = range_max - 0.1f;  // :493
  hipLaunchKernelGGL(ingest_laser_scan_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_ranges,
                     reinterpret_cast<const float2*>(h->d_trig), n, range_min, maxRangeForContainer, scale_to_map,
                     h->d_ingest, ingest_count_ptr(h));
  HIP_TRY(hipGetLastError());
  h->ingest_origo[0] = h->ingest_origo[1] = 0.0f;  // dataContainer.setOrigo(Vector2f::Zero()), :491
  return finish_ingest(h, out_pts_xy, out_n);
}

// shared tail of the two point-cloud entries
static int ingest_cloud(hsm_ctx* h, CloudIngestParams& P, const double tf_rows[12], float sqr_min, float sqr_max,
                        float z_min, float z_max, float scale_to_map, float* out_pts_xy, int* out_n,
                        float out_origo[2]) {
  for (int k = 0; k < 12; ++k) P.T[k] = tf_rows[k];
  P.sqr_min = sqr_min;
  P.sqr_max = sqr_max;
  P.z_min = z_min;
  P.z_max = z_max;
  P.scale = scale_to_map;
  P.out = h->d_ingest;
  P.out_n = ingest_count_ptr(h);
  hipLaunchKernelGGL(ingest_point_cloud_kernel, dim3(1), dim3(1024), 0, h->stream, P);
  HIP_TRY(hipGetLastError());
  // dataContainer.setOrigo(Eigen::Vector2f(laserPos.x(), laserPos.y()) * scaleToMap)  (:517)
  h->ingest_origo[0] = (float)tf_rows[3] * scale_to_map;
  h->ingest_origo[1] = (float)tf_rows[7] * scale_to_map;
  if (out_origo) {
    out_origo[0] = h->ingest_origo[0];
    out_origo[1] = h->ingest_origo[1];
  }
  return finish_ingest(h, out_pts_xy, out_n);
}

int hsm_ingest_point_cloud(hsm_ctx* h, const float* pts_xyz, int n, const double tf_rows[12], float sqr_laser_min_dist,
                           float sqr_laser_max_dist, float laser_z_min, float laser_z_max, float scale_to_map,
                           float* out_pts_xy, int* out_n, float out_origo[2]) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (n < 0 || (n > 0 && !pts_xyz) || !tf_rows || n > HSM_MAX_UPDATE_BEAMS)
    return fail(HSM_ERR_INVALID, "hsm_ingest_point_cloud: bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = ensure_ingest_capacity(h, n)) return rc;
  if (n > 0)
    HIP_TRY(hipMemcpyAsync(h->d_ranges, pts_xyz, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CloudIngestParams P{};
  P.pts_xyz = h->d_ranges;
  P.n = n;
  return ingest_cloud(h, P, tf_rows, sqr_laser_min_dist, sqr_laser_max_dist, laser_z_min, laser_z_max, scale_to_map,
                      out_pts_xy, out_n, out_origo);
}

int hsm_ingest_laser_scan_tf(hsm_ctx* h, const float* ranges, int n, float angle_min, float angle_increment,
                             float range_min, float range_max, double range_cutoff, const double tf_rows[12],
                             float sqr_laser_min_dist, float sqr_laser_max_dist, float laser_z_min, float laser_z_max,
                             float scale_to_map, float* out_pts_xy, int* out_n, float out_origo[2]) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (n < 0 || (n > 0 && !ranges) || !tf_rows || n > HSM_MAX_UPDATE_BEAMS)
    return fail(HSM_ERR_INVALID, "hsm_ingest_laser_scan_tf: bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = ensure_ingest_capacity(h, n)) return rc;
  if (h->trig_kind != 1 || h->trig_n != n || h->trig_a0 != angle_min || h->trig_inc != angle_increment) {
    // laser_geometry's unit vectors (getUnitVectors_): double cos/sin(angle_min + (double)i * angle_increment),
    // cached per sensor geometry there as well
    std::vector<double> t(2 * (size_t)n);
    const double a0 = angle_min, inc = angle_increment;
    for (int i = 0; i < n; ++i) {
      t[2 * i] = cos(a0 + (double)i * inc);
      t[2 * i + 1] = sin(a0 + (double)i * inc);
    }
    if (n > 0) HIP_TRY(hipMemcpy(h->d_trig, t.data(), (size_t)n * sizeof(double2), hipMemcpyHostToDevice));
    h->trig_kind = 1;
    h->trig_n = n;
    h->trig_a0 = angle_min;
    h->trig_inc = angle_increment;
  }
  if (n > 0) HIP_TRY(hipMemcpyAsync(h->d_ranges, ranges, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  CloudIngestParams P{};
  P.ranges = h->d_ranges;
  P.unit = reinterpret_cast<const double2*>(h->d_trig);
  P.n = n;
  P.range_min = range_min;
  P.range_cutoff = range_cutoff < 0 ? (double)range_max : range_cutoff;
  return ingest_cloud(h, P, tf_rows, sqr_laser_min_dist, sqr_laser_max_dist, laser_z_min, laser_z_max, scale_to_map,
                      out_pts_xy, out_n, out_origo);
}

int hsm_match_ingested(hsm_ctx* h, const float begin_world[3], float out_pose_world[3], float cov[9]) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (h->ingest_n < 0) return fail(HSM_ERR_INVALID, "hsm_match_ingested: no scan ingested");
  static const float dummy[2] = {0.0f, 0.0f};
  const float* hp = h->ingest_n > 0 ? h->h_ingest.data() : dummy;
  return match_impl(h, begin_world, hp, h->ingest_n, h->ingest_origo, out_pose_world, cov, nullptr, 0, h->d_ingest);
}

int hsm_update_by_ingested(hsm_ctx* h, const float pose_world[3]) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (!pose_world || h->ingest_n < 0) return fail(HSM_ERR_INVALID, "hsm_update_by_ingested: no scan ingested");
  std::lock_guard<std::mutex> lk(h->mu);
  static const float dummy[2] = {0.0f, 0.0f};
  const float* hp = h->ingest_n > 0 ? h->h_ingest.data() : dummy;
  return update_impl(h, pose_world, hp, h->ingest_n, h->ingest_origo, h->d_ingest);
}

static int ensure_batch_bytes(hsm_ctx* h, size_t need) {
  if (need <= h->d_batch_cap) return HSM_OK;
  if (h->d_batch) HIP_TRY(hipFree(h->d_batch));
  h->d_batch = nullptr;
  h->d_batch_cap = 0;
  HIP_TRY(hipMalloc(&h->d_batch, need));
  h->d_batch_cap = need;
  return HSM_OK;
}

static int score_states(hsm_ctx* h, int level, int batch, const float* states_map, const float* pts_xy, int n,
                        float* out_lh, float* out_residual, const char* who) {
  if (int rc = valid_level(h, level)) return rc;
  if (batch < 0 || n < 0 || (batch > 0 && (!states_map || !(out_lh || out_residual))) || (n > 0 && !pts_xy))
    return fail(HSM_ERR_INVALID, who);
  if (batch == 0) return HSM_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = ensure_scan_capacity(h->d_scan, h->d_scan_cap, (size_t)n)) return rc;
  if (n > 0) HIP_TRY(hipMemcpyAsync(h->d_scan, pts_xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  if (int rc = ensure_batch_bytes(h, (size_t)batch * 5 * sizeof(float))) return rc;
  float* d_states = (float*)h->d_batch;
  float* d_lh = d_states + 3 * (size_t)batch;
  float* d_res = d_lh + (size_t)batch;
  HIP_TRY(hipMemcpyAsync(d_states, states_map, (size_t)batch * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  const float factor = (float)(1.0 / pow(2.0, (double)level));
  const LevelView v = level_view(h->levels[level], factor, 1);
  const int grid = (batch + 3) / 4;
#define HSM_LAUNCH_LH(LAY, EX)                                                                                         \
  hipLaunchKernelGGL((likelihood_kernel<LAY, EX>), dim3(grid), dim3(256), 0, h->stream, v, d_states, batch, h->d_scan, \
                     n, factor, out_lh ? d_lh : nullptr, out_residual ? d_res : nullptr)
  if (h->layout == kLayoutPlane) {
    if (wants_exact(h)) HSM_LAUNCH_LH(kLayoutPlane, true); else HSM_LAUNCH_LH(kLayoutPlane, false);
  } else {
    if (wants_exact(h)) HSM_LAUNCH_LH(kLayoutQuad, true); else HSM_LAUNCH_LH(kLayoutQuad, false);
  }
#undef HSM_LAUNCH_LH
  HIP_TRY(hipGetLastError());
  if (out_lh) HIP_TRY(hipMemcpyAsync(out_lh, d_lh, (size_t)batch * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (out_residual)
    HIP_TRY(hipMemcpyAsync(out_residual, d_res, (size_t)batch * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

int hsm_likelihood_states(hsm_ctx* h, int level, int batch, const float* states_map, const float* pts_xy, int n,
                          float* out_lh) {
  return score_states(h, level, batch, states_map, pts_xy, n, out_lh, nullptr, "hsm_likelihood_states: bad argument");
}

int hsm_residual_states(hsm_ctx* h, int level, int batch, const float* states_map, const float* pts_xy, int n,
                        float* out_residual) {
  return score_states(h, level, batch, states_map, pts_xy, n, nullptr, out_residual,
                      "hsm_residual_states: bad argument");
}

int hsm_covariance_for_poses(hsm_ctx* h, int level, int batch, const float* poses_map, const float* pts_xy, int n,
                             float* out_cov_map, float* out_cov_world, float* out_lh7) {
  if (int rc = valid_level(h, level)) return rc;
  if (batch < 0 || n < 0 || (batch > 0 && (!poses_map || !(out_cov_map || out_cov_world || out_lh7))) ||
      (n > 0 && !pts_xy))
    return fail(HSM_ERR_INVALID, "hsm_covariance_for_poses: bad argument");
  if (batch == 0) return HSM_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = ensure_scan_capacity(h->d_scan, h->d_scan_cap, (size_t)n)) return rc;
  if (n > 0) HIP_TRY(hipMemcpyAsync(h->d_scan, pts_xy, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  if (int rc = ensure_batch_bytes(h, (size_t)batch * (3 + 9 + 9 + 7) * sizeof(float))) return rc;
  float* d_poses = (float*)h->d_batch;
  float* d_map = d_poses + 3 * (size_t)batch;
  float* d_world = d_map + 9 * (size_t)batch;
  float* d_lh7 = d_world + 9 * (size_t)batch;
  HIP_TRY(hipMemcpyAsync(d_poses, poses_map, (size_t)batch * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  const float factor = (float)(1.0 / pow(2.0, (double)level));
  const Level& Lv = h->levels[level];
  const LevelView v = level_view(Lv, factor, 1);
#define HSM_LAUNCH_COV(LAY, EX)                                                                                     \
  hipLaunchKernelGGL((pose_covariance_kernel<LAY, EX>), dim3(batch), dim3(448), 0, h->stream, v, d_poses, batch, \
                     h->d_scan, n, factor, Lv.cell_length, d_map, d_world, d_lh7)
  if (h->layout == kLayoutPlane) {
    if (wants_exact(h)) HSM_LAUNCH_COV(kLayoutPlane, true); else HSM_LAUNCH_COV(kLayoutPlane, false);
  } else {
    if (wants_exact(h)) HSM_LAUNCH_COV(kLayoutQuad, true); else HSM_LAUNCH_COV(kLayoutQuad, false);
  }
#undef HSM_LAUNCH_COV
  HIP_TRY(hipGetLastError());
  if (out_cov_map)
    HIP_TRY(hipMemcpyAsync(out_cov_map, d_map, (size_t)batch * 9 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (out_cov_world)
    HIP_TRY(hipMemcpyAsync(out_cov_world, d_world, (size_t)batch * 9 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (out_lh7)
    HIP_TRY(hipMemcpyAsync(out_lh7, d_lh7, (size_t)batch * 7 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

int hsm_ray_distances(hsm_ctx* h, int level, float origin_x, float origin_y, float resolution, int n,
                      const float* begin_world_xy, const float* end_world_xy, float* out_dist, float* out_hit_xy) {
  if (int rc = valid_level(h, level)) return rc;
  if (n < 0 || (n > 0 && (!begin_world_xy || !end_world_xy || !out_dist)) || !(resolution > 0.0f))
    return fail(HSM_ERR_INVALID, "hsm_ray_distances: bad argument");
  if (n == 0) return HSM_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  const size_t need = (size_t)n * 7 * sizeof(float);  // begin[2] end[2] dist[1] hit[2]
  if (need > h->d_batch_cap) {
    if (h->d_batch) HIP_TRY(hipFree(h->d_batch));
    h->d_batch = nullptr;
    h->d_batch_cap = 0;
    HIP_TRY(hipMalloc(&h->d_batch, need));
    h->d_batch_cap = need;
  }
  float* d = (float*)h->d_batch;
  RayQueryParams P;
  const Level& L = h->levels[level];
  P.logodds = L.d_logodds;
  P.sx = L.sx;
  P.sy = L.sy;
  P.origin_x = origin_x;
  P.origin_y = origin_y;
  P.scale = resolution;
  P.inv_scale = 1.0f / resolution;  // CoordinateTransformer::setTransforms, HectorMapTools.h:64
  P.begin_world = reinterpret_cast<const float2*>(d);
  P.end_world = reinterpret_cast<const float2*>(d + 2 * (size_t)n);
  P.out_hit = reinterpret_cast<float2*>(d + 4 * (size_t)n);
  P.out_dist = d + 6 * (size_t)n;
  P.n = n;
  HIP_TRY(hipMemcpyAsync(d, begin_world_xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(d + 2 * (size_t)n, end_world_xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if (out_hit_xy)  // in/out: rays without a hit keep the caller's values
    HIP_TRY(hipMemcpyAsync(d + 4 * (size_t)n, out_hit_xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(ray_distance_kernel, dim3((n + 3) / 4), dim3(256), 0, h->stream, P);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out_dist, P.out_dist, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (out_hit_xy)
    HIP_TRY(hipMemcpyAsync(out_hit_xy, P.out_hit, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

int hsm_occupancy_grid(hsm_ctx* h, int level, signed char* out) {
  if (int rc = valid_level(h, level)) return rc;
  if (!out) return fail(HSM_ERR_INVALID, "hsm_occupancy_grid: out is null");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  Level& L = h->levels[level];
  if (L.cells() > h->d_occ_cap) {
    (void)hipFree(h->d_occ);
    h->d_occ = nullptr;
    h->d_occ_cap = 0;
    HIP_TRY(hipMalloc((void**)&h->d_occ, L.cells()));
    h->d_occ_cap = L.cells();
  }
  hipLaunchKernelGGL(occupancy_grid_kernel, dim3(grid_for(L.cells() / 4)), dim3(256), 0, h->stream, L.d_logodds,
                     L.cells(), h->d_occ);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, h->d_occ, L.cells(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

int hsm_retain_scan(hsm_ctx* h, const float* pts_xy, int n, const float origo[2]) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  if (n < 0 || (n > 0 && !pts_xy)) return fail(HSM_ERR_INVALID, "hsm_retain_scan: bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->levels.size() > 1) {
    h->retained_pts.assign(pts_xy, pts_xy + 2 * (size_t)n);
    h->retained_origo[0] = origo ? origo[0] : 0.0f;
    h->retained_origo[1] = origo ? origo[1] : 0.0f;
    h->retained_valid = true;
    h->d_retained_current = false;
  }
  return HSM_OK;
}

}  // extern "C"

// One persistent host thread per replica beyond the first (replica 0 runs on the calling thread): a job slot guarded by
// a mutex + condition variable; threads live as long as the group, so a batched match costs no thread creation.
struct GroupWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = false, quit = false;
  int rc = HSM_OK;
  std::string err;
};

// RCCL, loaded on first use: the single-GPU library keeps its dependency set (HIP / HSA / libc), and a process that never
// gathers across devices never maps the 570 MB librccl.  In a process that has PyTorch-ROCm loaded the SONAME resolves to
// the librccl torch already brought in (one RCCL, one HIP runtime); elsewhere to /opt/rocm/lib.
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string error;
};

static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) {
      const char* e = dlerror();
      api.error = std::string("dlopen(librccl.so.1): ") + (e ? e : "not found");
      return;
    }
    bool ok = true;
    auto sym = [&](const char* n) -> void* {
      void* p = dlsym(api.lib, n);
      if (!p) {
        ok = false;
        api.error = std::string("librccl: missing symbol ") + n;
      }
      return p;
    };
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
    if (!ok) {
      dlclose(api.lib);
      api.lib = nullptr;
    }
  });
  return &api;
}

struct hsm_group {
  std::vector<hsm_ctx*> members;
  std::vector<std::unique_ptr<GroupWorker>> workers;  // workers[r - 1] serves replica r
  // device-resident gather (hsm_group_match_batch_device): per replica a result block on ITS device and an event
  std::vector<float*> d_pose, d_cov;
  std::vector<size_t> d_cap;  // scans
  std::vector<hipEvent_t> evt;
  // the gather itself: RCCL over the group's devices (one communicator per replica, ncclCommInitAll on first use), or
  // peer copies.  gather_pref = what was asked for (hsm_group_set_gather / env HSM_GROUP_GATHER), gather_mode = what runs.
  int gather_pref = HSM_GATHER_AUTO, gather_mode = HSM_GATHER_AUTO;
  bool force_p2p = false;  // hsm_group_debug_force_p2p: every shard, the root's too, through grouped ncclSend / ncclRecv
  std::vector<ncclComm_t> comms;
  std::vector<float*> d_all_pose, d_all_cov;  // all-gather receive blocks of the replicas other than the root
  std::vector<size_t> d_all_cap;              // floats of pose block (cov block: 3x)
  std::string gather_note;                    // why AUTO settled on peer copies, if it did
  // HSM_GATHER_DIRECT: one mailbox exchange per replica (pose_exchange.hip), re-made when the gathered row count changes
  std::vector<hsm_exchange*> xpose, xcov;
  size_t x_rows = 0;
  std::mutex mu;  // one group call at a time
};

#define NCCL_TRY(api, expr)                                                                     \
  do {                                                                                          \
    ncclResult_t r__ = (expr);                                                                  \
    if (r__ != ncclSuccess) {                                                                   \
      char b__[384];                                                                            \
      snprintf(b__, sizeof b__, "%s: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(r__) : "rccl error"); \
      return fail(HSM_ERR_HIP, b__);                                                            \
    }                                                                                           \
  } while (0)

// decide (once) how the group gathers: RCCL needs the library, distinct devices and a communicator per replica
static int group_ensure_gather(hsm_group* g) {
  if (g->gather_mode != HSM_GATHER_AUTO) return HSM_OK;
  const int R = (int)g->members.size();
  auto settle_peer = [&](const std::string& why) -> int {
    if (g->gather_pref == HSM_GATHER_RCCL) return fail(HSM_ERR_HIP, ("hsm_group: RCCL gather requested but unavailable: " + why).c_str());
    g->gather_note += why;
    g->gather_mode = HSM_GATHER_PEER;
    return HSM_OK;
  };
  if (g->gather_pref == HSM_GATHER_PEER) {
    g->gather_mode = HSM_GATHER_PEER;
    return HSM_OK;
  }
  std::vector<int> devs;
  for (hsm_ctx* h : g->members) devs.push_back(h->device);
  if (g->gather_pref == HSM_GATHER_AUTO || g->gather_pref == HSM_GATHER_DIRECT) {
    // the device-side exchange needs every replica's kernels to store into every other replica's HBM: the same device, or
    // peer access (xGMI on one node)
    std::string why;
    if (R > HSM_EXCHANGE_MAX_WORLD) why = "more replicas than HSM_EXCHANGE_MAX_WORLD";
    for (int a = 0; a < R && why.empty(); ++a)
      for (int b = 0; b < R && why.empty(); ++b) {
        if (devs[a] == devs[b]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devs[a], devs[b]) != hipSuccess || !can) {
          (void)hipGetLastError();
          why = "no peer access between devices " + std::to_string(devs[a]) + " and " + std::to_string(devs[b]);
        }
      }
    if (why.empty()) {
      g->gather_mode = HSM_GATHER_DIRECT;
      return HSM_OK;
    }
    if (g->gather_pref == HSM_GATHER_DIRECT) return fail(HSM_ERR_HIP, ("hsm_group: direct gather requested but unavailable: " + why).c_str());
    g->gather_note = "direct exchange unavailable (" + why + "); ";
  }
  for (int a = 0; a < R; ++a)
    for (int b = a + 1; b < R; ++b)
      if (devs[a] == devs[b]) return settle_peer("a device is listed more than once (one RCCL rank per device)");
  RcclApi* api = rccl_api();
  if (!api->lib) return settle_peer(api->error);
  g->comms.assign((size_t)R, nullptr);
  const ncclResult_t r = api->CommInitAll(g->comms.data(), R, devs.data());
  if (r != ncclSuccess) {
    g->comms.clear();
    return settle_peer(std::string("ncclCommInitAll: ") + api->GetErrorString(r));
  }
  g->gather_mode = HSM_GATHER_RCCL;
  return HSM_OK;
}

static void group_worker_main(GroupWorker* w) {
  std::unique_lock<std::mutex> lk(w->m);
  for (;;) {
    w->cv.wait(lk, [w] { return w->has_job || w->quit; });
    if (w->quit) return;
    std::function<int()> job = std::move(w->job);
    w->has_job = false;
    lk.unlock();
    const int rc = job();
    std::string err = rc != HSM_OK ? hsm_last_error() : "";  // thread-local text: carry it to the caller's thread
    lk.lock();
    w->rc = rc;
    w->err = std::move(err);
    w->done = true;
    w->cv.notify_all();
  }
}

// run fn(replica index) on every replica concurrently; first non-zero status wins
template <typename F>
static int group_parallel(hsm_group* g, F fn) {
  const int R = (int)g->members.size();
  for (int r = 1; r < R; ++r) {
    GroupWorker* w = g->workers[(size_t)r - 1].get();
    std::lock_guard<std::mutex> lk(w->m);
    w->job = [fn, r]() -> int { return fn(r); };
    w->has_job = true;
    w->done = false;
    w->cv.notify_all();
  }
  int rc0 = fn(0);
  int rc_out = rc0;
  std::string err_out = rc0 != HSM_OK ? std::string(hsm_last_error()) : std::string();
  for (int r = 1; r < R; ++r) {
    GroupWorker* w = g->workers[(size_t)r - 1].get();
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [w] { return w->done; });
    if (w->rc != HSM_OK && rc_out == HSM_OK) {
      rc_out = w->rc;
      err_out = w->err;
    }
  }
  return rc_out == HSM_OK ? HSM_OK : fail(rc_out, err_out.c_str());
}


extern "C" {

int hsm_group_create(float map_resolution, int size_x, int size_y, unsigned levels, float start_x, float start_y,
                     const int* devices, int n_devices, hsm_group** out) {
  if (!out || !devices || n_devices < 1) return fail(HSM_ERR_INVALID, "hsm_group_create: bad argument");
  *out = nullptr;
  hsm_group* g = new hsm_group();
  for (int i = 0; i < n_devices; ++i) {
    hsm_opts o;
    o.device = devices[i];
    o.layout = HSM_LAYOUT_AUTO;
    o.waves_per_scan = 0;
    hsm_ctx* h = nullptr;
    const int rc = hsm_create(map_resolution, size_x, size_y, levels, start_x, start_y, &o, &h);
    if (rc != HSM_OK) {
      hsm_group_destroy(g);
      return rc;
    }
    g->members.push_back(h);
  }
  for (int i = 1; i < n_devices; ++i) {
    g->workers.emplace_back(new GroupWorker());
    GroupWorker* w = g->workers.back().get();
    w->th = std::thread(group_worker_main, w);
  }
  g->d_pose.assign((size_t)n_devices, nullptr);
  g->d_cov.assign((size_t)n_devices, nullptr);
  g->d_cap.assign((size_t)n_devices, 0);
  g->evt.assign((size_t)n_devices, nullptr);
  g->d_all_pose.assign((size_t)n_devices, nullptr);
  g->d_all_cov.assign((size_t)n_devices, nullptr);
  g->d_all_cap.assign((size_t)n_devices, 0);
  if (const char* env = getenv("HSM_GROUP_GATHER")) {
    if (strcmp(env, "rccl") == 0) g->gather_pref = HSM_GATHER_RCCL;
    else if (strcmp(env, "peer") == 0) g->gather_pref = HSM_GATHER_PEER;
    else if (strcmp(env, "direct") == 0) g->gather_pref = HSM_GATHER_DIRECT;
    else if (strcmp(env, "auto") != 0) {
      hsm_group_destroy(g);
      return fail(HSM_ERR_INVALID, "hsm_group_create: HSM_GROUP_GATHER must be one of auto, direct, rccl, peer");
    }
  }
  *out = g;
  return HSM_OK;
}

int hsm_group_set_gather(hsm_group* g, int mode) {
  if (!g) return fail(HSM_ERR_INVALID, "null group");
  if (mode != HSM_GATHER_AUTO && mode != HSM_GATHER_PEER && mode != HSM_GATHER_RCCL && mode != HSM_GATHER_DIRECT)
    return fail(HSM_ERR_INVALID, "hsm_group_set_gather: unknown mode");
  std::lock_guard<std::mutex> glk(g->mu);
  g->gather_pref = mode;
  g->gather_note.clear();
  if (mode == HSM_GATHER_PEER) {
    g->gather_mode = HSM_GATHER_PEER;
    return HSM_OK;
  }
  if (mode == HSM_GATHER_RCCL && !g->comms.empty()) {  // communicators, once made, are kept and reused
    g->gather_mode = HSM_GATHER_RCCL;
    return HSM_OK;
  }
  g->gather_mode = HSM_GATHER_AUTO;  // decide again
  return mode == HSM_GATHER_AUTO ? HSM_OK : group_ensure_gather(g);
}

int hsm_group_debug_force_p2p(hsm_group* g, int on) {
  if (!g) return fail(HSM_ERR_INVALID, "null group");
  std::lock_guard<std::mutex> glk(g->mu);
  g->force_p2p = on != 0;
  return HSM_OK;
}

int hsm_group_gather_mode(hsm_group* g) {
  if (!g) return HSM_GATHER_AUTO;
  std::lock_guard<std::mutex> glk(g->mu);
  if (group_ensure_gather(g) != HSM_OK) return HSM_GATHER_AUTO;
  return g->gather_mode;
}

const char* hsm_group_gather_note(const hsm_group* g) { return g ? g->gather_note.c_str() : ""; }

void hsm_group_destroy(hsm_group* g) {
  if (!g) return;
  for (auto& w : g->workers) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->quit = true;
      w->cv.notify_all();
    }
    if (w->th.joinable()) w->th.join();
  }
  if (!g->comms.empty()) {
    for (hsm_ctx* h : g->members) (void)hsm_synchronize(h);
    RcclApi* api = rccl_api();
    for (ncclComm_t c : g->comms)
      if (c && api->CommDestroy) (void)api->CommDestroy(c);
  }
  if (!g->xpose.empty() || !g->xcov.empty()) {
    for (hsm_ctx* h : g->members) (void)hsm_synchronize(h);
    for (hsm_exchange* x : g->xpose) hsm_exchange_destroy(x);
    for (hsm_exchange* x : g->xcov) hsm_exchange_destroy(x);
  }
  TeardownLog log_, *log = &log_;  // (as hsm_destroy: name the first failing call, leave no error behind for the next caller)
  for (size_t r = 0; r < g->members.size(); ++r) {
    if (g->members[r]) TEARDOWN(log, hipSetDevice(g->members[r]->device));
    if (r < g->d_all_pose.size()) {
      TEARDOWN(log, hipFree(g->d_all_pose[r]));
      TEARDOWN(log, hipFree(g->d_all_cov[r]));
    }
    if (r < g->d_pose.size()) {
      TEARDOWN(log, hipFree(g->d_pose[r]));
      TEARDOWN(log, hipFree(g->d_cov[r]));
      if (g->evt[r]) TEARDOWN(log, hipEventDestroy(g->evt[r]));
    }
  }
  for (hsm_ctx* h : g->members) hsm_destroy(h);
  delete g;
  if (!log_.first.empty()) {
    g_last_error = "hsm_group_destroy: " + log_.first;
    (void)hipGetLastError();
  }
}

int hsm_group_size(const hsm_group* g) { return g ? (int)g->members.size() : 0; }

hsm_ctx* hsm_group_member(hsm_group* g, int i) {
  return (g && i >= 0 && i < (int)g->members.size()) ? g->members[i] : nullptr;
}

int hsm_group_set_update_factors(hsm_group* g, float free_factor, float occupied_factor) {
  if (!g) return fail(HSM_ERR_INVALID, "null group");
  for (hsm_ctx* h : g->members) {
    if (int rc = hsm_set_update_factor_free(h, free_factor)) return rc;
    if (int rc = hsm_set_update_factor_occupied(h, occupied_factor)) return rc;
  }
  return HSM_OK;
}

int hsm_group_process_scan(hsm_group* g, const float hint_world[3], const float* pts_xy, int n, const float origo[2],
                           int do_update, float out_pose_world[3], float cov[9]) {
  if (!g || g->members.empty()) return fail(HSM_ERR_INVALID, "null group");
  std::lock_guard<std::mutex> glk(g->mu);
  if (int rc = hsm_match(g->members[0], hint_world, pts_xy, n, origo, out_pose_world, cov)) return rc;
  if (!do_update) return HSM_OK;
  return group_parallel(g, [&](int r) -> int {
    hsm_ctx* h = g->members[r];
    if (r != 0)
      if (int rc = hsm_retain_scan(h, pts_xy, n, origo)) return rc;
    return hsm_update_by_scan(h, out_pose_world, pts_xy, n, origo);
  });
}

int hsm_group_match_batch_device(hsm_group* g, const int* counts, const float* const* d_begin_world,
                                 const float* const* d_pts_xy, const int* const* d_scan_offsets, int shared_n, int root,
                                 float* d_out_pose_all, float* d_out_cov_all) {
  if (!g || g->members.empty()) return fail(HSM_ERR_INVALID, "null group");
  const int R = (int)g->members.size();
  if (!counts || !d_begin_world || !d_pts_xy || !d_out_pose_all || root < 0 || root >= R)
    return fail(HSM_ERR_INVALID, "hsm_group_match_batch_device: bad argument");
  std::lock_guard<std::mutex> glk(g->mu);
  std::vector<size_t> first((size_t)R + 1, 0);
  for (int r = 0; r < R; ++r) {
    if (counts[r] < 0 || (counts[r] > 0 && (!d_begin_world[r] || !d_pts_xy[r])))
      return fail(HSM_ERR_INVALID, "hsm_group_match_batch_device: bad shard");
    first[(size_t)r + 1] = first[(size_t)r] + (size_t)counts[r];
  }
  const int root_dev = g->members[(size_t)root]->device;
  if (int rc = group_ensure_gather(g)) return rc;
  const bool rccl = g->gather_mode == HSM_GATHER_RCCL;
  const bool direct = g->gather_mode == HSM_GATHER_DIRECT;
  const size_t total = first[(size_t)R];
  if (direct && total > 0) {
    // Device-side exchange: one mailbox per replica for [total, 3] (+ one for [total, 9]), made when the gathered row count
    // changes (a particle filter keeps its particle count; anything else pays a re-allocation here)
    const bool want_cov = d_out_cov_all != nullptr;
    if (g->x_rows != total || g->xpose.empty() || (want_cov && g->xcov.empty())) {
      for (hsm_ctx* h : g->members)
        if (int rc0 = hsm_synchronize(h)) return rc0;
      const bool remake_pose = g->x_rows != total || g->xpose.empty();
      auto make = [&](std::vector<hsm_exchange*>& xs, int cols) -> int {
        for (hsm_exchange* x : xs) hsm_exchange_destroy(x);
        xs.assign((size_t)R, nullptr);
        for (int r = 0; r < R; ++r)
          if (int rc0 = hsm_exchange_create(g->members[(size_t)r]->device, r, R, (int)total, cols, 2, &xs[(size_t)r])) return rc0;
        for (int r = 0; r < R; ++r)
          if (int rc0 = hsm_exchange_connect_local(xs[(size_t)r], xs.data())) return rc0;
        return HSM_OK;
      };
      if (remake_pose) {
        if (int rc0 = make(g->xpose, 3)) return rc0;
        for (hsm_exchange* x : g->xcov) hsm_exchange_destroy(x);  // (shaped for the old row count)
        g->xcov.clear();
      }
      if (want_cov && g->xcov.empty())
        if (int rc0 = make(g->xcov, 9)) return rc0;
      g->x_rows = total;
    }
  }
  bool equal = counts[0] > 0;  // ncclAllGather wants the same count from every rank
  for (int r = 1; r < R; ++r) equal = equal && counts[r] == counts[0];
  const bool self_send = rccl && g->force_p2p;  // test hook: the send / receive form for every shard, the root's own included
  if (self_send) equal = false;
  // every replica: match its shard on its own stream.  Peer gather: push the poses (and H) to the root's device with a peer
  // copy on the same stream -- 12 (+36) bytes per scan over xGMI, no host staging, no host wait.  RCCL gather: the
  // collective is queued below, behind the match, on the same streams.
  int rc = group_parallel(g, [&](int r) -> int {
    hsm_ctx* h = g->members[(size_t)r];
    const size_t n = (size_t)counts[r];
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc2 = select_device(h)) return rc2;
    if (!g->evt[(size_t)r]) HIP_TRY(hipEventCreateWithFlags(&g->evt[(size_t)r], hipEventDisableTiming));
    if (((rccl && equal) || direct) && r != root && total * 3 > g->d_all_cap[(size_t)r]) {  // all-gather receive blocks of a non-root replica
      (void)hipFree(g->d_all_pose[(size_t)r]);
      (void)hipFree(g->d_all_cov[(size_t)r]);
      g->d_all_pose[(size_t)r] = g->d_all_cov[(size_t)r] = nullptr;
      g->d_all_cap[(size_t)r] = 0;
      HIP_TRY(hipMalloc((void**)&g->d_all_pose[(size_t)r], total * 3 * sizeof(float)));
      HIP_TRY(hipMalloc((void**)&g->d_all_cov[(size_t)r], total * 9 * sizeof(float)));
      g->d_all_cap[(size_t)r] = total * 3;
    }
    if (n > 0) {
      if (n > g->d_cap[(size_t)r]) {
        (void)hipFree(g->d_pose[(size_t)r]);
        (void)hipFree(g->d_cov[(size_t)r]);
        g->d_pose[(size_t)r] = g->d_cov[(size_t)r] = nullptr;
        g->d_cap[(size_t)r] = 0;
        HIP_TRY(hipMalloc((void**)&g->d_pose[(size_t)r], n * 3 * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&g->d_cov[(size_t)r], n * 9 * sizeof(float)));
        g->d_cap[(size_t)r] = n;
      }
      if (int rc2 = match_batch_device_nolock(h, (int)n, d_begin_world[r], d_pts_xy[r],
                                              d_scan_offsets ? d_scan_offsets[r] : nullptr, shared_n, g->d_pose[(size_t)r],
                                              d_out_cov_all ? g->d_cov[(size_t)r] : nullptr, h->stream))
        return rc2;
      if (direct) {
        // (below, also for a replica without scans: every replica posts every epoch)
      } else if (!rccl || (r == root && !equal && !self_send)) {  // (RCCL send/recv gather: the root's own shard is a local copy)
        HIP_TRY(hipMemcpyPeerAsync(d_out_pose_all + 3 * first[(size_t)r], root_dev, g->d_pose[(size_t)r], h->device,
                                   n * 3 * sizeof(float), h->stream));
        if (d_out_cov_all)
          HIP_TRY(hipMemcpyPeerAsync(d_out_cov_all + 9 * first[(size_t)r], root_dev, g->d_cov[(size_t)r], h->device,
                                     n * 9 * sizeof(float), h->stream));
      }
    }
    if (direct && total > 0) {
      // ONE launch on this replica's stream, behind its match: store the shard's rows into every replica's mailbox and
      // unpack all shards' rows as they arrive -- the root into the caller's arrays, the others into blocks the group
      // keeps (every replica holds all poses afterwards: hsm_group_gathered).  No collective, no event, no host wait.
      if (int rc2 = hsm_exchange_post_wait(g->xpose[(size_t)r], g->d_pose[(size_t)r], (int)first[(size_t)r], (int)n, 0,
                                           r == root ? d_out_pose_all : g->d_all_pose[(size_t)r], h->stream))
        return rc2;
      if (d_out_cov_all)
        if (int rc2 = hsm_exchange_post_wait(g->xcov[(size_t)r], g->d_cov[(size_t)r], (int)first[(size_t)r], (int)n, 0,
                                             r == root ? d_out_cov_all : g->d_all_cov[(size_t)r], h->stream))
          return rc2;
      return HSM_OK;
    }
    if (!rccl) HIP_TRY(hipEventRecord(g->evt[(size_t)r], h->stream));
    return HSM_OK;
  });
  if (rc != HSM_OK) return rc;
  if (direct) return HSM_OK;
  if (rccl) {
    // ONE grouped collective over the group's communicators, each rank's part on its replica's stream (behind its match):
    // equal shards -> ncclAllGather of [B/G, 3] (+ [B/G, 9]); the root receives straight into the caller's arrays, the
    // other replicas into blocks the group keeps (every replica then holds all poses: hsm_group_gathered).  Unequal
    // shards -> the same gather as grouped ncclSend / ncclRecv to the root.  The collective itself orders the root's
    // stream behind every shard.
    RcclApi* api = rccl_api();
    NCCL_TRY(api, api->GroupStart());
    ncclResult_t nr = ncclSuccess;
    for (int r = 0; r < R && nr == ncclSuccess; ++r) {
      hsm_ctx* h = g->members[(size_t)r];
      const size_t n = (size_t)counts[r];
      if (equal) {
        nr = api->AllGather(g->d_pose[(size_t)r], r == root ? d_out_pose_all : g->d_all_pose[(size_t)r], n * 3, ncclFloat,
                            g->comms[(size_t)r], h->stream);
        if (nr == ncclSuccess && d_out_cov_all)
          nr = api->AllGather(g->d_cov[(size_t)r], r == root ? d_out_cov_all : g->d_all_cov[(size_t)r], n * 9, ncclFloat,
                              g->comms[(size_t)r], h->stream);
      } else if ((r != root || self_send) && n > 0) {
        hsm_ctx* hr = g->members[(size_t)root];
        nr = api->Send(g->d_pose[(size_t)r], n * 3, ncclFloat, root, g->comms[(size_t)r], h->stream);
        if (nr == ncclSuccess)
          nr = api->Recv(d_out_pose_all + 3 * first[(size_t)r], n * 3, ncclFloat, r, g->comms[(size_t)root], hr->stream);
        if (nr == ncclSuccess && d_out_cov_all) {
          nr = api->Send(g->d_cov[(size_t)r], n * 9, ncclFloat, root, g->comms[(size_t)r], h->stream);
          if (nr == ncclSuccess)
            nr = api->Recv(d_out_cov_all + 9 * first[(size_t)r], n * 9, ncclFloat, r, g->comms[(size_t)root], hr->stream);
        }
      }
    }
    const ncclResult_t ne = api->GroupEnd();
    if (nr != ncclSuccess) NCCL_TRY(api, nr);
    NCCL_TRY(api, ne);
    return HSM_OK;
  }
  // the root's stream waits for every shard: work queued on it afterwards (and hsm_synchronize on the root member) sees
  // the complete gather
  hsm_ctx* hr = g->members[(size_t)root];
  std::lock_guard<std::mutex> lk(hr->mu);
  if (int rc2 = select_device(hr)) return rc2;
  for (int r = 0; r < R; ++r)
    if (r != root) HIP_TRY(hipStreamWaitEvent(hr->stream, g->evt[(size_t)r], 0));
  return HSM_OK;
}

const float* hsm_group_gathered(hsm_group* g, int replica, int want_cov) {
  if (!g || replica < 0 || replica >= (int)g->d_all_pose.size()) return nullptr;
  return want_cov ? g->d_all_cov[(size_t)replica] : g->d_all_pose[(size_t)replica];
}

int hsm_group_synchronize(hsm_group* g) {
  if (!g) return fail(HSM_ERR_INVALID, "null group");
  for (hsm_ctx* h : g->members)
    if (int rc = hsm_synchronize(h)) return rc;
  for (hsm_exchange* x : g->xpose)  // a gather whose rows did not all arrive says so here
    if (int rc = hsm_exchange_status(x)) return rc;
  for (hsm_exchange* x : g->xcov)
    if (int rc = hsm_exchange_status(x)) return rc;
  return HSM_OK;
}

// THE partitioning of a batch over G replicas (contiguous shards): the first total % world shards hold one
// scan more.  One rule for both transports -- hsm_group_* (one process, a worker thread per device) and
// hector_slam_amd/sharding.py (one process per device under torch.distributed, which calls this function).
int hsm_shard_bounds(int total, int rank, int world, int* begin, int* end) {
  if (total < 0 || world <= 0 || rank < 0 || rank >= world || !begin || !end) return fail(HSM_ERR_INVALID, "hsm_shard_bounds: bad argument");
  const int base = total / world, rem = total % world;
  *begin = rank * base + (rank < rem ? rank : rem);
  *end = *begin + base + (rank < rem ? 1 : 0);
  return HSM_OK;
}

int hsm_group_match_batch(hsm_group* g, int batch, const float* begin_world, const float* pts_xy,
                          const int* scan_offsets, int shared_n, float* out_pose, float* out_cov) {
  if (!g || g->members.empty()) return fail(HSM_ERR_INVALID, "null group");
  if (batch < 0 || !begin_world || !out_pose) return fail(HSM_ERR_INVALID, "hsm_group_match_batch: bad argument");
  const int R = (int)g->members.size();
  std::lock_guard<std::mutex> glk(g->mu);
  return group_parallel(g, [&](int r) -> int {
    int b = 0, e = 0;
    if (int rc = hsm_shard_bounds(batch, r, R, &b, &e)) return rc;
    if (e == b) return HSM_OK;
    if (!scan_offsets)  // pose hypotheses of ONE shared scan
      return hsm_match_batch(g->members[r], e - b, begin_world + 3 * (size_t)b, pts_xy, nullptr, shared_n,
                             out_pose + 3 * (size_t)b, out_cov ? out_cov + 9 * (size_t)b : nullptr);
    std::vector<int> offs((size_t)(e - b) + 1);  // CSR offsets rebased to the shard
    for (int i = b; i <= e; ++i) offs[(size_t)(i - b)] = scan_offsets[i] - scan_offsets[b];
    return hsm_match_batch(g->members[r], e - b, begin_world + 3 * (size_t)b, pts_xy + 2 * (size_t)scan_offsets[b],
                           offs.data(), 0, out_pose + 3 * (size_t)b, out_cov ? out_cov + 9 * (size_t)b : nullptr);
  });
}

int hsm_level_info(const hsm_ctx* h, int level, int* sx, int* sy, float* cell, float* scale) {
  if (int rc = valid_level(h, level)) return rc;
  const Level& L = h->levels[level];
  if (sx) *sx = L.sx;
  if (sy) *sy = L.sy;
  if (cell) *cell = L.cell_length;
  if (scale) *scale = L.scale_to_map;
  return HSM_OK;
}
int hsm_map_coords_pose(const hsm_ctx* h, int level, const float w[3], float m[3]) {
  if (int rc = valid_level(h, level)) return rc;
  affine_apply_host(h->levels[level].mapTworld, w[0], w[1], m[0], m[1]);
  m[2] = w[2];
  return HSM_OK;
}
int hsm_world_coords_pose(const hsm_ctx* h, int level, const float m[3], float w[3]) {
  if (int rc = valid_level(h, level)) return rc;
  affine_apply_host(h->levels[level].worldTmap, m[0], m[1], w[0], w[1]);
  w[2] = m[2];
  return HSM_OK;
}
int hsm_update_index(const hsm_ctx* h, int level) {
  if (valid_level(h, level)) return -1;
  std::lock_guard<std::mutex> lk(h->mu);  // read by the facade's publisher thread while the scan thread updates
  return h->levels[level].last_update_index;
}

int hsm_download_level(hsm_ctx* h, int level, float* logodds, int* update_index) {
  if (int rc = valid_level(h, level)) return rc;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  Level& L = h->levels[level];
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (logodds) HIP_TRY(hipMemcpy(logodds, L.d_logodds, L.cells() * sizeof(float), hipMemcpyDeviceToHost));
  if (update_index)
    HIP_TRY(hipMemcpy(update_index, L.d_update_index, L.cells() * sizeof(int), hipMemcpyDeviceToHost));
  return HSM_OK;
}
int hsm_upload_level(hsm_ctx* h, int level, const float* logodds, const int* update_index) {
  if (int rc = valid_level(h, level)) return rc;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  Level& L = h->levels[level];
  if (int rc = order_after_foreign_match(h)) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (logodds) HIP_TRY(hipMemcpy(L.d_logodds, logodds, L.cells() * sizeof(float), hipMemcpyHostToDevice));
  if (update_index)
    HIP_TRY(hipMemcpy(L.d_update_index, update_index, L.cells() * sizeof(int), hipMemcpyHostToDevice));
  if (int rc = rebuild_probability(h, L)) return rc;
  L.dirty[0] = L.dirty[1] = 0;
  L.dirty[2] = L.sx - 1;
  L.dirty[3] = L.sy - 1;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}
int hsm_download_rows(hsm_ctx* h, int level, int y0, int y1, float* rows) {
  if (int rc = valid_level(h, level)) return rc;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  Level& L = h->levels[level];
  if (y0 < 0 || y1 > L.sy || y0 > y1 || !rows) return fail(HSM_ERR_INVALID, "hsm_download_rows: bad row range");
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (y1 > y0)
    HIP_TRY(hipMemcpy(rows, L.d_logodds + (size_t)y0 * L.sx, (size_t)(y1 - y0) * L.sx * sizeof(float),
                      hipMemcpyDeviceToHost));
  return HSM_OK;
}
int hsm_download_cells(hsm_ctx* h, int level, int x0, int y0, int x1, int y1, void* dst_cells, int dst_pitch_cells) {
  if (int rc = valid_level(h, level)) return rc;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  Level& L = h->levels[level];
  if (x0 < 0 || y0 < 0 || x1 >= L.sx || y1 >= L.sy || x1 < x0 || y1 < y0 || !dst_cells || dst_pitch_cells < x1 - x0 + 1)
    return fail(HSM_ERR_INVALID, "hsm_download_cells: bad rectangle");
  const int w = x1 - x0 + 1, hgt = y1 - y0 + 1;
  const size_t need = (size_t)w * hgt * 8;
  if (need > h->d_cells_cap) {
    if (h->d_cells) HIP_TRY(hipFree(h->d_cells));
    h->d_cells = nullptr;
    h->d_cells_cap = 0;
    HIP_TRY(hipMalloc(&h->d_cells, need + need / 2));
    h->d_cells_cap = need + need / 2;
  }
  hipLaunchKernelGGL(pack_cells_kernel, dim3(grid_for((size_t)w * hgt)), dim3(256), 0, h->stream, level_rw(L), x0, y0, w,
                     hgt, reinterpret_cast<int2*>(h->d_cells));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy2DAsync(dst_cells, (size_t)dst_pitch_cells * 8, h->d_cells, (size_t)w * 8, (size_t)w * 8, hgt,
                           hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}
int hsm_last_update_bbox(const hsm_ctx* h, int level, int bbox[4]) {
  if (int rc = valid_level(h, level)) return rc;
  for (int i = 0; i < 4; ++i) bbox[i] = h->levels[level].bbox[i];
  return HSM_OK;
}
int hsm_take_dirty_bbox(hsm_ctx* h, int level, int bbox[4]) {
  if (int rc = valid_level(h, level)) return rc;
  if (!bbox) return fail(HSM_ERR_INVALID, "hsm_take_dirty_bbox: bbox is null");
  std::lock_guard<std::mutex> lk(h->mu);
  Level& L = h->levels[level];
  for (int i = 0; i < 4; ++i) bbox[i] = L.dirty[i];
  L.dirty[0] = L.dirty[1] = 0;
  L.dirty[2] = L.dirty[3] = -1;
  return HSM_OK;
}
int hsm_download_prob(hsm_ctx* h, int level, float* prob) {
  if (int rc = valid_level(h, level)) return rc;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  Level& L = h->levels[level];
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(prob, L.d_prob, L.cells() * sizeof(float), hipMemcpyDeviceToHost));
  return HSM_OK;
}

int hsm_hessian_derivs(hsm_ctx* h, int level, const float pose_map[3], const float* pts, int n, float H[9],
                       float dTr[3]) {
  if (int rc = valid_level(h, level)) return rc;
  if (!pose_map || n < 0 || (n > 0 && !pts) || !H || !dTr) return fail(HSM_ERR_INVALID, "bad argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = ensure_scan_capacity(h->d_scan, h->d_scan_cap, (size_t)n)) return rc;
  if (n > 0) HIP_TRY(hipMemcpyAsync(h->d_scan, pts, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  const LevelView v = level_view(h->levels[level], 1.0f, 1);
  float* d_out = h->d_small + 16;
  if (wants_exact(h)) {
    if (h->layout == kLayoutPlane)
      hipLaunchKernelGGL((gn_eval_kernel<kLayoutPlane, true>), dim3(1), dim3(1024), 0, h->stream, v, h->d_scan, n,
                         pose_map[0], pose_map[1], pose_map[2], d_out);
    else
      hipLaunchKernelGGL((gn_eval_kernel<kLayoutQuad, true>), dim3(1), dim3(1024), 0, h->stream, v, h->d_scan, n,
                         pose_map[0], pose_map[1], pose_map[2], d_out);
  } else if (h->layout == kLayoutPlane)
    hipLaunchKernelGGL((gn_eval_kernel<kLayoutPlane>), dim3(1), dim3(1024), 0, h->stream, v, h->d_scan, n,
                       pose_map[0], pose_map[1], pose_map[2], d_out);
  else
    hipLaunchKernelGGL((gn_eval_kernel<kLayoutQuad>), dim3(1), dim3(1024), 0, h->stream, v, h->d_scan, n,
                       pose_map[0], pose_map[1], pose_map[2], d_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(h->h_small + 16, d_out, 12 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 9; ++i) H[i] = h->h_small[16 + i];
  for (int i = 0; i < 3; ++i) dTr[i] = h->h_small[25 + i];
  return HSM_OK;
}

int hsm_eval_beams(hsm_ctx* h, int level, const float pose_map[3], const float* pts, int n, float* out4) {
  if (int rc = valid_level(h, level)) return rc;
  if (!pose_map || n < 0 || (n > 0 && (!pts || !out4))) return fail(HSM_ERR_INVALID, "bad argument");
  if (n == 0) return HSM_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  if (int rc = ensure_scan_capacity(h->d_scan, h->d_scan_cap, (size_t)n * 3)) return rc;  // pts + float4 out
  HIP_TRY(hipMemcpyAsync(h->d_scan, pts, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
  float4* d_out = reinterpret_cast<float4*>(h->d_scan + (((size_t)n + 1) & ~(size_t)1));
  const LevelView v = level_view(h->levels[level], 1.0f, 1);
  const int grid = (n + 255) / 256;
  if (h->layout == kLayoutPlane)
    hipLaunchKernelGGL((gn_beam_terms_kernel<kLayoutPlane>), dim3(grid), dim3(256), 0, h->stream, v, h->d_scan, n,
                       pose_map[0], pose_map[1], pose_map[2], d_out);
  else
    hipLaunchKernelGGL((gn_beam_terms_kernel<kLayoutQuad>), dim3(grid), dim3(256), 0, h->stream, v, h->d_scan, n,
                       pose_map[0], pose_map[1], pose_map[2], d_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out4, d_out, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return HSM_OK;
}

int hsm_debug_set_coop_barrier(hsm_ctx* h, unsigned value) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(h->d_partials + 2 * 64 * 12, &value, sizeof value, hipMemcpyHostToDevice));
  h->coop_bar_base = value;
  return HSM_OK;
}

int hsm_debug_set_coop_mute(hsm_ctx* h, int block_plus_one) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  h->coop_mute_block = block_plus_one;
  return HSM_OK;
}

int hsm_debug_spec_stats(hsm_ctx* h, int enable, unsigned long long out[4]) {
  if (!h) return fail(HSM_ERR_INVALID, "null context");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (out) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (h->d_spec_stats) HIP_TRY(hipMemcpy(out, h->d_spec_stats, sizeof(SpecStats), hipMemcpyDeviceToHost));
  }
  if (enable && !h->d_spec_stats) HIP_TRY(hipMalloc((void**)&h->d_spec_stats, sizeof(SpecStats)));
  if (h->d_spec_stats) HIP_TRY(hipMemset(h->d_spec_stats, 0, sizeof(SpecStats)));
  if (!enable && h->d_spec_stats) {
    HIP_TRY(hipFree(h->d_spec_stats));
    h->d_spec_stats = nullptr;
  }
  return HSM_OK;
}

int hsm_debug_coop_fallbacks(hsm_ctx* h) {
  if (!h) return 0;
  std::lock_guard<std::mutex> lk(h->mu);
  return (int)h->coop_fallbacks;
}

int hsm_debug_marks_nonzero(hsm_ctx* h, int level, unsigned long long out[2]) {
  if (int rc = valid_level(h, level)) return rc;
  if (!out) return fail(HSM_ERR_INVALID, "hsm_debug_marks_nonzero: out is null");
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  Level& L = h->levels[level];
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 2 * sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), h->stream);
  if (e == hipSuccess) {
    const size_t nb = (mark_plane_bytes(L.sx, L.sy)) / 4, nw = (L.cells() + 31) / 32 + 1;
    hipLaunchKernelGGL(count_nonzero_words_kernel, dim3(grid_for(nb)), dim3(256), 0, h->stream,
                       reinterpret_cast<const unsigned int*>(L.d_free_bytes), nb, d);
    hipLaunchKernelGGL(count_nonzero_words_kernel, dim3(grid_for(nw)), dim3(256), 0, h->stream, L.d_occ_bits, nw, d + 1);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(HSM_ERR_HIP, "hsm_debug_marks_nonzero", e);
  return HSM_OK;
}

int hsm_debug_set_update_serial(hsm_ctx* h, int level, unsigned serial) {
  if (int rc = valid_level(h, level)) return rc;
  if (serial > kSerialMax) return fail(HSM_ERR_INVALID, "hsm_debug_set_update_serial: serial exceeds the key generation field");
  std::lock_guard<std::mutex> lk(h->mu);
  h->levels[level].serial = serial;
  return HSM_OK;
}

int hsm_debug_expf(hsm_ctx* h, int n, const float* x, float* out_exp, float* out_prob) {
  if (!h || n < 0 || (n > 0 && (!x || !out_exp || !out_prob))) return fail(HSM_ERR_INVALID, "hsm_debug_expf: bad argument");
  if (n == 0) return HSM_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  float* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 3 * (size_t)n * sizeof(float)));
  hipError_t e = hipMemcpyAsync(d, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(expf_debug_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, d, n, d + n, d + 2 * (size_t)n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_exp, d + n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out_prob, d + 2 * (size_t)n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(HSM_ERR_HIP, "hsm_debug_expf", e);
  return HSM_OK;
}

int hsm_debug_sincos(hsm_ctx* h, int n, const float* x, float* s, float* c) {
  if (!h || n < 0 || (n > 0 && (!x || !s || !c))) return fail(HSM_ERR_INVALID, "hsm_debug_sincos: bad argument");
  if (n == 0) return HSM_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  if (int rc = select_device(h)) return rc;
  float* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 3 * (size_t)n * sizeof(float)));
  hipError_t e = hipMemcpyAsync(d, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(sincos_debug_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, d, n, d + n, d + 2 * (size_t)n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(s, d + n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c, d + 2 * (size_t)n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(HSM_ERR_HIP, "hsm_debug_sincos", e);
  return HSM_OK;
}

}  // extern "C"
