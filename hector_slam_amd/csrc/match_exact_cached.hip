// match_exact_cached.hip -- launches of the exact-order texel-cache batch forms (gn_match_exact.h): the headline kernel of
// configs[2] / configs[3] and its chain-wavefront forms.  A translation unit of its own: these twelve instantiations are a third
// of the library's compile time, and the kernel is the one that gets edited.
#include "gn_match_exact.h"
#include "hsm_ctx.h"

namespace hsm_host {
namespace {

#define HIP_TRY HSM_HIP_TRY

// beams-per-lane register budget: the smallest instantiated BPL that holds max_n beams in the
// team's VGPRs (0 = stream the endpoints from memory every GN step)
// HSM_PARITY_EXACT: the exact-order form of the general kernel (endpoints streamed, no texel cache)
template <int NS, int BPL, int BPC = BPL, bool CW = false, bool PROBE = false>
int launch_match_exact_cached(hsm_ctx* h, MatchParams P, hipStream_t stream) {
  int grid = (P.batch + NS - 1) / NS;
  const int block = 64 * (NS + (CW ? 1 : 0));
  if (CW) P.xp.world = 0;  // (the chain-wavefront forms do not carry it: the caller queues the stand-alone exchange kernel)
  if (P.xp.world > 0) {    // this launch carries the pose exchange: every scan posts its pose, the workgroups behind the matcher's own unpack
    P.xp.match_blocks = grid;
    grid += P.xp.wait_blocks;
    h->fused_exchange_done = true;
  }
  // workgroup -> XCD mapping: this form runs best with one contiguous eighth of the batch per XCD on every map size (2048^2
  // headline: 57.5 us against 58.3 with the fast form's chunks of 16 workgroups dealt in turn; chunks of 8 / 32: 58.4;
  // profiles/r04/exact_kernel_param_sweep.txt) -- its rounds are paced by barriers and chain jobs, not by how long a scan's
  // gathers take, so the load balancing the chunks buy the fast form is not needed and the compacter L2 footprint wins.
  // env HSM_XCD_CHUNK_EXACT=n restores chunks of n workgroups.
  P.xcd_chunk = h->xcd_chunk_exact > 0 ? (h->xcd_chunk_exact * 4 / NS > 0 ? h->xcd_chunk_exact * 4 / NS : 1) : 0;
  hipLaunchKernelGGL((gn_match_exact_cached_kernel<NS, BPL, BPC, CW, PROBE>), dim3(grid), dim3(block), 0, stream, P);
  HIP_TRY(hipGetLastError());
  h->last_kernel = CW ? "gn_match_exact_cached_kernel (chain wavefront)" : "gn_match_exact_cached_kernel";
  h->last_cfg[0] = h->layout;
  h->last_cfg[1] = 1;
  h->last_cfg[2] = block;
  h->last_cfg[3] = grid;
  h->last_cfg[4] = BPL;
  h->last_cfg[5] = 1;
  return HSM_OK;
}

// the texel-cache exact forms of launch_match_exact, by scan length and by how many workgroups the launch leaves a CU
}  // namespace

int launch_match_exact_cached_forms(hsm_ctx* h, const MatchParams& P, int max_n, hipStream_t stream) {
  const int per_lane = (max_n + 63) / 64;
  // A launch that leaves every CU at most THREE workgroups takes the chain-wavefront form (gn_match_exact.h, CW): a fifth
  // wavefront per workgroup runs the chain jobs, so a round lasts max(job, production) instead of job + production --
  // 36 us against 52 for a level-0 batch of up to 2048 scans, 49 against 57 at 3072 (profiles/r05/README.md 9).  Not
  // beyond: the dispatcher places a workgroup only where EVERY SIMD has room for ceil(waves / 4) of its wavefronts
  // (tools/study/ubench_wg_placement.hip), the fourth five-wavefront workgroup of a CU waits for a whole workgroup to
  // retire, and at four per CU both forms deliver the same ~70 scans per us anyway.
  const int groups = (P.batch + 3) / 4;
  // ... and a map that outgrows the L2s (4096^2: 136 us with six cached rows against 128.5 with fifteen, at 3072 scans) keeps
  // round 3's form at three workgroups per CU; up to two per CU the chain-wavefront form has the full texel cache as well
  const bool cw2 = h->exact_chain_wave && groups <= 2 * h->compute_units;
  const bool cw = cw2 || (h->exact_chain_wave && groups <= 3 * h->compute_units && h->levels[0].cells() <= ((size_t)1 << 23));
  if (per_lane <= 5) return cw ? launch_match_exact_cached<4, 5, 5, true>(h, P, stream) : launch_match_exact_cached<4, 5>(h, P, stream);
  if (per_lane <= 9) return cw ? launch_match_exact_cached<4, 9, 9, true>(h, P, stream) : launch_match_exact_cached<4, 9>(h, P, stream);
  // (a round loop that leaves behind the longest scan's last row costs the 17-row form 8 % on full-length scans -- sixteen
  // exit edges --; a 13-row instantiation costs compile time only: a batch of 720-beam scans runs 13 rounds instead of 17)
  if (per_lane <= 13) {
    if (cw2) return launch_match_exact_cached<4, 13, 13, true>(h, P, stream);
    return cw ? launch_match_exact_cached<4, 13, HSM_XBPC_CW + 1, true>(h, P, stream) : launch_match_exact_cached<4, 13>(h, P, stream);
  }
  if (cw2) return launch_match_exact_cached<4, 17, HSM_XBPC, true>(h, P, stream);
  if (cw) return launch_match_exact_cached<4, 17, HSM_XBPC_CW, true>(h, P, stream);
  // four workgroups per CU: the balanced schedule (13 cached rows) where level 0 fits the L2s, else round 3's (15 cached rows: the
  // gathers of a map that misses the L2 cost more than the schedule gains)
  if (HSM_XBPC_MAIN != HSM_XBPC && h->levels[0].cells() > ((size_t)1 << 23)) return launch_match_exact_cached<4, 17, HSM_XBPC>(h, P, stream);
  // (hsm_set_clock_probe: the headline form has an instantiation that carries the stamps)
  if (P.clock_probe != nullptr) return launch_match_exact_cached<4, 17, HSM_XBPC_MAIN, false, true>(h, P, stream);
  return launch_match_exact_cached<4, 17, HSM_XBPC_MAIN>(h, P, stream);
}

}  // namespace hsm_host
