// match_teams.hip -- launches of the team forms of the matcher (gn_match.h: gn_match_kernel on 1 .. 16 wavefronts per scan,
// either summation order) and of the fast texel-cache forms (gn_match_cached_kernel).  About eighty instantiations: a
// translation unit of its own so that an edit elsewhere does not rebuild them.
#include "gn_match.h"
#include "hsm_ctx.h"

namespace hsm {
// ---- hsm_set_batch_order: the batch in Morton order of its start poses ---------------------------------------------------------
// Scans that sit next to each other in a launch run on the same XCD at the same time and share the texel lines they touch in its
// L2; a batch in an order that does not follow the map loses that (profiles/r06: +5 % on the 2048^2 map, +22 % on the 4096^2
// pyramid).  ONE workgroup: counting sort of the batch by the Morton code of the 64 x 64 level-0 tile its start pose lies in
// (4096 bins in LDS); the order inside a tile is whatever the LDS atomics make it -- any order gives the same results, a scan's
// result depends on nothing but the scan.  perm[slot] = scan.
__device__ __forceinline__ unsigned part1by1_6(unsigned v) {  // 6 bits -> every other bit
  v &= 0x3fu;
  v = (v | (v << 4)) & 0x30fu;
  v = (v | (v << 2)) & 0x333u;
  v = (v | (v << 1)) & 0x555u;
  return v;
}

// a position in bin `key` for every valid lane.  A batch that already follows the map has 64 equal keys per wavefront, which the LDS
// would serialise: a wavefront whose lanes all hold ONE key takes one atomic between them; otherwise an atomic per lane (no loop: the
// eight calls of a thread stay independent instruction streams)
__device__ __forceinline__ int bin_take(int* bins, int key, bool valid, int lane) {
  const unsigned long long todo = __ballot(valid);
  if (todo == 0ull) return 0;
  const int k = __builtin_amdgcn_readfirstlane(key);  // (the first active lane's: lanes beyond the batch sit at the end)
  const unsigned long long m = __ballot(valid && key == k);
  int pos = 0;
  if (m == todo && __builtin_amdgcn_readfirstlane((int)valid) != 0) {  // one key
    if (valid) {
      int base = 0;
      const int first = __ffsll((long long)m) - 1;
      if (lane == first) base = atomicAdd(&bins[k], (int)__popcll(m));
      pos = __shfl(base, first) + (int)__popcll(m & ((1ull << lane) - 1ull));
    }
  } else if (valid) {
    pos = atomicAdd(&bins[key], 1);
  }
  return pos;
}

// detect != 0 (HSM_ORDER_AUTO): a batch that follows the map already -- its tile changes between neighbours number no more than a
// few times its distinct tiles -- keeps its own order (the identity permutation): the order inside a tile would only get worse.
__global__ void __launch_bounds__(1024) batch_order_kernel(Affine2 mapTworld, int tile_shift, const float* __restrict__ begin_world, int batch,
                                                           int* __restrict__ perm, int detect) {
  constexpr int KPT = 8;  // scans per thread and pass: their start poses are loaded together, not one dependent round trip each
  __shared__ int bins[4096];
  __shared__ int part[1024];
  __shared__ int changes, tiles;
  const int tid = (int)threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 1024) bins[i] = 0;
  if (tid == 0) changes = 0, tiles = 0;
  __syncthreads();
  auto keys_of = [&](int first, int (&key)[KPT]) {
    float x[KPT], y[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
      const int i = first + u * 1024 + tid;
      x[u] = i < batch ? begin_world[3 * i + 0] : 0.0f;
      y[u] = i < batch ? begin_world[3 * i + 1] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
      float mx, my;
      affine_apply(mapTworld, x[u], y[u], mx, my);
      // (a NaN or far-away start estimate: any tile will do)
      const int cx = mx == mx ? (int)fminf(fmaxf(mx, 0.0f), 1.0e6f) : 0, cy = my == my ? (int)fminf(fmaxf(my, 0.0f), 1.0e6f) : 0;
      const int tx = min(cx >> tile_shift, 63), ty = min(cy >> tile_shift, 63);
      key[u] = (int)(part1by1_6((unsigned)tx) | (part1by1_6((unsigned)ty) << 1));
    }
  };
  int key[KPT];
  for (int first = 0; first < batch; first += 1024 * KPT) {
    keys_of(first, key);
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
      const bool valid = first + u * 1024 + tid < batch;
      bin_take(bins, key[u], valid, lane);
      if (detect) {  // neighbours in the batch are neighbours in the wavefront (the first lane's left neighbour is not looked at)
        const int left = __shfl_up(key[u], 1);
        const unsigned long long m = __ballot(valid && lane > 0 && key[u] != left);
        if (lane == 0 && m != 0ull) atomicAdd(&changes, (int)__popcll(m));
      }
    }
  }
  __syncthreads();
  const int c0 = bins[4 * tid], c1 = bins[4 * tid + 1], c2 = bins[4 * tid + 2], c3 = bins[4 * tid + 3];
  if (detect) {
    const int wave_tiles = (int)(__popcll(__ballot(c0 != 0)) + __popcll(__ballot(c1 != 0)) + __popcll(__ballot(c2 != 0)) + __popcll(__ballot(c3 != 0)));
    if (lane == 0 && wave_tiles) atomicAdd(&tiles, wave_tiles);
    __syncthreads();
    if (changes <= 4 * tiles + 16) {  // (workgroup-uniform)
      for (int i = tid; i < batch; i += 1024) perm[i] = i;
      return;
    }
  }
  // exclusive scan of the 1024 partial counts: inside the wavefront by DPP-free shuffles, then over the 16 wavefront totals
  const int mine = c0 + c1 + c2 + c3;
  int inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(inc, d);
    if (lane >= d) inc += v;
  }
  if (lane == 63) part[tid >> 6] = inc;
  __syncthreads();
  int wave_base = 0;
  for (int w = 0; w < (tid >> 6); ++w) wave_base += part[w];
  const int base = wave_base + inc - mine;
  bins[4 * tid] = base, bins[4 * tid + 1] = base + c0, bins[4 * tid + 2] = base + c0 + c1, bins[4 * tid + 3] = base + c0 + c1 + c2;
  __syncthreads();
  for (int first = 0; first < batch; first += 1024 * KPT) {
    if (batch > 1024 * KPT) keys_of(first, key);  // (a batch of one pass still holds its keys)
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
      const int i = first + u * 1024 + tid;
      const int pos = bin_take(bins, key[u], i < batch, lane);
      if (i < batch) perm[pos] = i;
    }
  }
}

}  // namespace hsm

namespace hsm_host {

#define HIP_TRY HSM_HIP_TRY

int ensure_batch_perm(hsm_ctx* h, MatchParams& P, hipStream_t stream) {
  if (P.perm != nullptr || P.begin_world == nullptr || P.batch < h->batch_order_min) return HSM_OK;
  const bool automatic = h->batch_order == HSM_ORDER_AUTO;
  if (h->batch_order != HSM_ORDER_MORTON && !(automatic && h->levels[0].cells() > ((size_t)1 << 23))) return HSM_OK;
  hsm_ctx::PermBuf* pb = nullptr;
  for (hsm_ctx::PermBuf& b : h->perm_bufs)
    if (b.s == stream) pb = &b;
  if (!pb) {
    if (h->perm_bufs.size() >= 8) return HSM_OK;  // (a ninth stream keeps the caller's order)
    h->perm_bufs.push_back({stream, nullptr, 0, 0, 0});
    pb = &h->perm_bufs.back();
  }
  if (pb->cap < (size_t)P.batch) {
    // (no allocation while the caller captures this stream into a graph: such a launch keeps the caller's order)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return HSM_OK;
    if (pb->d) HIP_TRY(hipFree(pb->d));  // (hipFree waits for the device: no launch still reads it)
    pb->d = nullptr, pb->cap = 0, pb->batch = 0;
    const size_t cap = ((size_t)P.batch + 4095) / 4096 * 4096;
    HIP_TRY(hipMalloc((void**)&pb->d, cap * sizeof(int)));
    pb->cap = cap;
  }
  if (pb->batch == P.batch && pb->used < h->batch_order_refresh) {  // the permutation of an earlier launch of this stream
    ++pb->used;
    P.perm = pb->d;
    h->last_sorted = true;
    return HSM_OK;
  }
  // 64 tiles span the longer edge of level 0
  const Level& L0 = h->levels[0];
  int shift = 0;
  while ((64 << shift) < (L0.sx > L0.sy ? L0.sx : L0.sy)) ++shift;
  hipLaunchKernelGGL(batch_order_kernel, dim3(1), dim3(1024), 0, stream, P.lv[0].mapTworld, shift, P.begin_world, P.batch, pb->d, automatic ? 1 : 0);
  HIP_TRY(hipGetLastError());
  pb->batch = P.batch, pb->used = 1;
  P.perm = pb->d;
  h->last_sorted = true;
  return HSM_OK;
}

namespace {

template <int WPS, int SPB, int BPL>
int launch_match_t(hsm_ctx* h, const MatchParams& P_in, hipStream_t stream) {
  const MatchParams& P = P_in;
  const int block = 64 * WPS * SPB;
  const int grid = (P.batch + SPB - 1) / SPB;
  if constexpr (WPS == 1 && (BPL == 9 || BPL == 17)) {
    // throughput launches of long scans: the texel-cache form (gn_match.h)
    if (h->texel_cache && P.begin_world && !P.trace) {
      MatchParams P = P_in;  // (hsm_set_batch_order: the launch takes its scans through a permutation)
      if (int rc = ensure_batch_perm(h, P, stream)) return rc;
      if (h->layout == kLayoutQuad && h->relaxed)
        hipLaunchKernelGGL((gn_match_cached_kernel<SPB, BPL, kLayoutQuad, 1, true>), dim3(grid), dim3(block), 0, stream, P);
      else if (h->layout == kLayoutQuad)
        hipLaunchKernelGGL((gn_match_cached_kernel<SPB, BPL, kLayoutQuad>), dim3(grid), dim3(block), 0, stream, P);
      else
        hipLaunchKernelGGL((gn_match_cached_kernel<SPB, BPL, kLayoutPlane>), dim3(grid), dim3(block), 0, stream, P);
      HIP_TRY(hipGetLastError());
      h->last_cfg[0] = h->layout;
      h->last_cfg[1] = WPS;
      h->last_cfg[2] = block;
      h->last_cfg[3] = grid;
      h->last_cfg[4] = BPL;
      h->last_cfg[5] = 1;
      h->last_kernel = "gn_match_cached_kernel";
      return HSM_OK;
    }
  }
  h->last_cfg[5] = 0;
  h->last_kernel = "gn_match_kernel";
  if (h->layout == kLayoutPlane)
    hipLaunchKernelGGL((gn_match_kernel<WPS, SPB, kLayoutPlane, BPL>), dim3(grid), dim3(block), 0, stream, P);
  else
    hipLaunchKernelGGL((gn_match_kernel<WPS, SPB, kLayoutQuad, BPL>), dim3(grid), dim3(block), 0, stream, P);
  HIP_TRY(hipGetLastError());
  h->last_cfg[0] = h->layout;
  h->last_cfg[1] = WPS;
  h->last_cfg[2] = block;
  h->last_cfg[3] = grid;
  h->last_cfg[4] = BPL;
  return HSM_OK;
}

template <int WPS, int SPB>
int launch_match_exact(hsm_ctx* h, const MatchParams& P_in, int max_n, hipStream_t stream) {
  MatchParams P = P_in;
  // throughput launches of the quad layout: every wavefront a producer with the texel cache, four scans per workgroup, one
  // 36-lane chain job per round behind the round's barrier (gn_match_exact.h).  Measured against round 2's producer /
  // chain-wavefront form (profiles/r03/README.md): 66-69 vs 92 us on the 2048^2 headline batch, 141-143 vs 199 us on the
  // 3-level batch, 156-162 vs 291 us on the 4096^2 pyramid.  Scans longer than 17 beams per lane stream their tail rows.
  if (WPS == 1 && P.begin_world && !P.trace && h->layout == kLayoutQuad && h->bpl_override != 0 && h->exact_cached) {
    if (int rc = ensure_batch_perm(h, P, stream)) return rc;  // (hsm_set_batch_order)
    // More than one generation of workgroups (four per CU) with a remainder that the chain-wavefront form takes: the whole
    // generations go out in round 3's form, the remainder behind them in its own launch -- 5000 scans: 57 + 36 us instead of the
    // 104 a single launch takes (its last, part-filled generation runs ~47 us in the rotating-owner form).
    const int groups = (P.batch + 3) / 4, full = 4 * h->compute_units, rest = groups % full;
    if (h->exact_chain_wave && h->exact_split_tail && groups > full && rest > 0 &&
        (rest <= 2 * h->compute_units || (rest <= 3 * h->compute_units && h->levels[0].cells() <= ((size_t)1 << 23)))) {
      MatchParams A = P, B = P;
      A.batch = (groups - rest) * 4;
      B.batch = P.batch - A.batch;
      if (P.perm) {  // (a permuted batch: the second launch takes the rest of the permutation, its scan indices stay absolute)
        B.perm = P.perm + A.batch;
      } else {
        B.begin_world = P.begin_world + 3 * (size_t)A.batch;
        if (P.offsets) B.offsets = P.offsets + A.batch;  // (absolute offsets into pts: the pointer moves, pts stays)
        B.out_pose = P.out_pose + 3 * (size_t)A.batch;
        if (P.out_cov) B.out_cov = P.out_cov + 9 * (size_t)A.batch;
      }
      B.clock_probe = nullptr;  // (scan 0's probe belongs to the first launch)
      // (a launch that carries the pose exchange: the part-filled last generation runs in a chain-wavefront form, which does not --
      // so the whole step is left to the stand-alone exchange kernel behind both launches)
      A.xp.world = 0;
      B.xp.world = 0;
      if (int rc = launch_match_exact_cached_forms(h, A, max_n, stream)) return rc;
      const int grid_a = h->last_cfg[3];
      if (int rc = launch_match_exact_cached_forms(h, B, max_n, stream)) return rc;
      h->last_cfg[2] = 256;  // (hsm_last_launch_config describes the first launch; its grid counts both)
      h->last_cfg[3] += grid_a;
      h->last_kernel = "gn_match_exact_cached_kernel + its chain-wavefront form for the last, part-filled generation";
      return HSM_OK;
    }
    return launch_match_exact_cached_forms(h, P, max_n, stream);
  }
#if defined(HSM_EXPERIMENTS)
  // round 2's exact batch form: producer wavefronts + chain wavefronts per workgroup (gn_match.h), env HSM_EXACT_CACHED=0.
  // Measured (profiles/r02/README.md): 108 vs 122 us on the 2048^2 headline batch with the <7,1> shape, 92-97 us with <8,2>
  // and two gathers in flight.  Not in the default library since round 4 (the texel-cache form above serves every quad
  // batch; the plane layout takes the one-wavefront exact form below).
  if (WPS == 1 && P.begin_world && !P.trace && h->exact_batch_form &&
      (h->exact_batch_form == 2 || h->levels[0].cells() <= ((size_t)1 << 23))) {
    const auto worst_cu = [&](int per_wg) { return ((P.batch + per_wg - 1) / per_wg + 255) / 256 * per_wg; };
    int per_wg = worst_cu(8) < worst_cu(kExactScans) ? 8 : kExactScans;
    if (h->exact_shape == 7 || h->exact_shape == 8) per_wg = h->exact_shape;
    const int grid = (P.batch + per_wg - 1) / per_wg, block = per_wg == 8 ? 64 * 10 : 64 * (kExactScans + 1);
    if (per_wg == 8) {
      if (h->layout == kLayoutPlane)
        hipLaunchKernelGGL((gn_match_exact_batch_kernel<kLayoutPlane, 8, 2>), dim3(grid), dim3(block), 0, stream, P);
      else
        hipLaunchKernelGGL((gn_match_exact_batch_kernel<kLayoutQuad, 8, 2>), dim3(grid), dim3(block), 0, stream, P);
    } else if (h->layout == kLayoutPlane) {
      hipLaunchKernelGGL((gn_match_exact_batch_kernel<kLayoutPlane>), dim3(grid), dim3(block), 0, stream, P);
    } else {
      hipLaunchKernelGGL((gn_match_exact_batch_kernel<kLayoutQuad>), dim3(grid), dim3(block), 0, stream, P);
    }
    HIP_TRY(hipGetLastError());
    h->last_cfg[0] = h->layout;
    h->last_cfg[1] = 1;
    h->last_cfg[2] = block;
    h->last_cfg[3] = grid;
    h->last_cfg[4] = 0;
    h->last_cfg[5] = 0;
    return HSM_OK;
  }
#endif
  const int block = 64 * WPS * SPB;
  const int grid = (P.batch + SPB - 1) / SPB;
  if (h->layout == kLayoutPlane)
    hipLaunchKernelGGL((gn_match_kernel<WPS, SPB, kLayoutPlane, 0, true>), dim3(grid), dim3(block), 0, stream, P);
  else
    hipLaunchKernelGGL((gn_match_kernel<WPS, SPB, kLayoutQuad, 0, true>), dim3(grid), dim3(block), 0, stream, P);
  HIP_TRY(hipGetLastError());
  h->last_cfg[0] = h->layout;
  h->last_cfg[1] = WPS;
  h->last_cfg[2] = block;
  h->last_cfg[3] = grid;
  h->last_cfg[4] = 0;
  h->last_cfg[5] = 0;
  h->last_kernel = "gn_match_kernel (exact order)";
  return HSM_OK;
}

template <int WPS, int SPB>
int launch_match_w(hsm_ctx* h, const MatchParams& P, int max_n, hipStream_t stream, bool exact) {
  if (exact) return launch_match_exact<WPS, SPB>(h, P, max_n, stream);
  const int per_lane = (max_n + 64 * WPS - 1) / (64 * WPS);
  if (h->bpl_override == 0 || per_lane > 17) return launch_match_t<WPS, SPB, 0>(h, P, stream);
  // (two beams per lane: only one-wavefront teams get there by themselves -- choose_wps keeps ~5 beams per lane -- so wider
  // teams, reachable through an explicit waves_per_scan only, share the three-beam instantiation)
  if constexpr (WPS == 1)
    if (per_lane <= 2) return launch_match_t<WPS, SPB, 2>(h, P, stream);
  if (per_lane <= 3) return launch_match_t<WPS, SPB, 3>(h, P, stream);
  if (per_lane <= 5) return launch_match_t<WPS, SPB, 5>(h, P, stream);
  if (per_lane <= 9) return launch_match_t<WPS, SPB, 9>(h, P, stream);
  return launch_match_t<WPS, SPB, 17>(h, P, stream);
}

}  // namespace

int launch_match_by_width(hsm_ctx* h, const MatchParams& P, int max_n, hipStream_t stream, bool exact, int wps) {
  switch (wps) {
    case 1: {
      // maps whose touched region outgrows the L2s: EIGHT consecutive scans per workgroup instead of four -- with the
      // per-beam workgroup barrier (MatchParams::wg_sync) eight waves share the texel lines in the CU's L1 (4096^2
      // pyramid: 132.8 -> 129.1 us; 16 per workgroup: 133 us; no effect on the 2048^2 workloads, which keep four)
      const int per_lane = (max_n + 63) / 64;
      if (h->spb_large == 8 && h->levels[0].cells() > ((size_t)1 << 23) && !exact && h->texel_cache && P.begin_world &&
          !P.trace && h->bpl_override != 0 && per_lane > 5 && per_lane <= 17)
        return per_lane <= 9 ? launch_match_t<1, 8, 9>(h, P, stream) : launch_match_t<1, 8, 17>(h, P, stream);
      return launch_match_w<1, 4>(h, P, max_n, stream, exact);
    }
    case 2: {
#if defined(HSM_EXPERIMENTS)
      // experimental (HSM_CACHED_WPS2=1, explicit waves_per_scan = 2): the texel-cache form on a PAIR of waves per scan
      // -- nine beams per lane, five waves per SIMD, 1.6 generations of waves for a 4096-scan launch (gn_match.h)
      const int per_lane = (max_n + 127) / 128;
      if (h->cached_wps2 && !exact && h->texel_cache && P.begin_world && !P.trace && h->bpl_override != 0 &&
          h->layout == kLayoutQuad && per_lane > 0 && per_lane <= 9) {
        hipLaunchKernelGGL((gn_match_cached_kernel<1, 9, kLayoutQuad, 2>), dim3(P.batch), dim3(128), 0, stream, P);
        HIP_TRY(hipGetLastError());
        h->last_cfg[0] = h->layout;
        h->last_cfg[1] = 2;
        h->last_cfg[2] = 128;
        h->last_cfg[3] = P.batch;
        h->last_cfg[4] = 9;
        h->last_cfg[5] = 1;
        return HSM_OK;
      }
#endif
      return launch_match_w<2, 1>(h, P, max_n, stream, exact);
    }
    case 4: return launch_match_w<4, 1>(h, P, max_n, stream, exact);
    case 8: return launch_match_w<8, 1>(h, P, max_n, stream, exact);
    default: return launch_match_w<16, 1>(h, P, max_n, stream, exact);
  }
}

}  // namespace hsm_host
