// hsm_ctx.h -- the context of libhector_mi355.so as its translation units see it: the pyramid level, hsm_ctx, and the launch
// entry points the units implement for each other.  Internal (not installed): the public boundary is include/hector_mi355/capi.h.
//   hector_mi355.hip        host runtime + C ABI, update / node-row / probe kernels, the one-workgroup-per-scan matcher forms
//   match_exact_cached.hip  the exact-order texel-cache batch forms (gn_match_exact.h: the headline kernel and its chain-wavefront forms)
//   match_teams.hip         the team forms (gn_match_kernel, 1..16 wavefronts per scan) and the fast texel-cache forms
//   pose_exchange.hip       the device-side gather of sharded results
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "gn_match.h"
#include "hector_mi355/capi.h"
#include "hsm_host.h"

namespace hsm {
struct BeamRec;  // map_update.h (its kernels are not templates: only hector_mi355.hip includes that header)
}

namespace hsm_host {

using namespace hsm;

// d_small / h_small layout (floats): [0,3) begin pose | [3,6) out pose | [6,15) out cov |
// [16,28) eval H,dTr | [kTraceOff, kTraceOff + 12 * max steps) per-step trace
constexpr int kTraceOff = 64;
constexpr int kDoneFlagOff = 32;  // one word of the pinned block: single-scan completion sequence number
constexpr int kErrFlagOff = 33;   // the next word: receives that number when the cooperative matcher's exchange timed out
constexpr int kMaxTraceSteps = 6 + 4 * (HSM_MAX_LEVELS - 1);
constexpr int kSmallFloats = kTraceOff + 12 * kMaxTraceSteps;

struct Level {
  int sx = 0, sy = 0;
  float cell_length = 0.f, scale_to_map = 0.f;
  float limx = 0.f, limy = 0.f;
  Affine2 mapTworld{}, worldTmap{};
  // device planes
  float* d_logodds = nullptr;
  int* d_update_index = nullptr;
  float* d_prob = nullptr;
  float4* d_quad = nullptr;
  unsigned int* d_key_free = nullptr;
  unsigned int* d_key_occ = nullptr;
  unsigned int* d_occ_bits = nullptr;
  unsigned char* d_free_bytes = nullptr;  // dense scans: crossed-cell byte map in the key_free tiling (map_update.h)
  // GridMapLogOddsFunctions (GridMapLogOdds.h:200-203)
  float log_odds_free = 0.f, log_odds_occ = 0.f;
  // OccGridMapBase counters / GridMapBase::lastUpdateIndex
  int curr_update_index = 0, curr_mark_occ = -1, curr_mark_free = -1, last_update_index = -1;
  unsigned int serial = 0;  // key-plane generation (map_update.h)
  int bbox[4] = {0, 0, -1, -1};   // cell box touched by the last update
  int dirty[4] = {0, 0, -1, -1};  // union of those boxes since hsm_take_dirty_bbox was last called
  int key_rows[2] = {0, -1};      // rows that carry keys of the current key generation (union of the boxes since the planes were last cleared)
  bool marks_pending = false;     // a mark pass was queued on this level and its apply pass has not been (scrub_marks)
  size_t cells() const { return (size_t)sx * sy; }
  int tiles_x() const { return (sx + 3) / 4; }
  int quad_texels() const {
#if HSM_QUAD_TILE
    return tiles_x() * ((sy + 1) / 2) * 8;
#else
    return sx * sy;
#endif
  }
};


}  // namespace hsm_host

using hsm_host::Level;
using hsm::BeamRec;
using hsm::SpecStats;
using hsm::kLayoutQuad;

struct hsm_ctx {
  int device = 0;
  int layout = kLayoutQuad;
  int wps_override = 0;
  // updateByScan returns when its kernels are QUEUED (env HSM_ASYNC_UPDATE=0: wait for them): everything
  // that reads the map afterwards is ordered behind them on `stream`.  Host endpoints are staged in one of
  // two pinned blocks, each guarded by the event of the update that last read it.
  bool texel_cache = true;          // env HSM_TEXEL_CACHE=0: plain gn_match_kernel for throughput launches too
  // ordering between the context's stream (updates) and caller-owned streams (hsm_match_batch_device):
  // per caller stream the update epoch it has been ordered behind, and whether it may still run a match
  struct ForeignStream {
    hipStream_t s;
    unsigned long long ordered_epoch;
    bool pending;
  };
  std::vector<ForeignStream> foreign;
  unsigned long long upd_epoch = 1;
  hipEvent_t evt_updates = nullptr, evt_foreign = nullptr;
  bool async_update = true;
  int update_zero_copy_max = 4096;  // env HSM_UPDATE_ZEROCOPY_MAX
  int merged_mark_max = 4096;       // scans below this take the one-launch mark pass (env HSM_MERGED_MARK_MAX, 0 = never)
  int scatter_texels_max = 1 << 30; // quad layout: scans below this write the texels from the apply pass (env HSM_SCATTER_TEXELS_MAX, 0 = never)
  BeamRec* d_beam_recs = nullptr;   // dense scans: per-beam records of all levels (map_update.h BeamRec), [levels][cap]
  size_t beam_recs_cap = 0;         // beams per level
  float2* h_upd_pinned[2] = {nullptr, nullptr};
  size_t h_upd_cap[2] = {0, 0};
  hipEvent_t upd_evt[2] = {nullptr, nullptr};
  bool upd_busy[2] = {false, false};
  int upd_slot = 0;
  bool spin_wait = true;       // single-scan matches: poll the kernel's completion word (env HSM_SPIN_WAIT=0: off)
  unsigned done_seq = 0;
  std::vector<hsm_host::Level> levels;
  mutable std::mutex mu;
  hipStream_t stream = nullptr;
  // single-scan staging (device) + pinned result
  float2* d_scan = nullptr;
  size_t d_scan_cap = 0;
  float* d_small = nullptr;   // begin pose[3] | out pose[3] | out cov[9] | eval[12]
  float* h_small = nullptr;   // pinned mirror of d_small
  // retained scan = MapRepMultiMap::dataContainers (level-0 units; scaled by 2^-l on use)
  std::vector<float> retained_pts;
  float retained_origo[2] = {0.f, 0.f};
  bool retained_valid = false;  // false until the first match (reference: empty containers)
  float2* d_retained = nullptr;
  size_t d_retained_cap = 0;
  bool d_retained_current = false;
  // the upload of a dense scan for matchData runs on its own stream, into the OTHER of two device buffers, from a pinned
  // staging block: it overlaps the update kernels still queued on `stream` (which read the buffer of the scan before)
  // instead of waiting behind them; the match kernel waits for the copy's event (stage_scan_overlapped)
  float2* d_retained_alt = nullptr;
  size_t d_retained_alt_cap = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_evt = nullptr;
  float2* h_copy_pinned = nullptr;
  size_t h_copy_pinned_cap = 0;
  bool overlap_upload = true;  // env HSM_OVERLAP_UPLOAD=0: the copy is queued on `stream` as before
  bool queued_update = false;  // an asynchronous updateByScan was queued on `stream` since the host last saw it drained
  // batch staging for the host-pointer convenience entry
  void* d_batch = nullptr;
  size_t d_batch_cap = 0;
  // ... and for its shared-scan form (pose hypotheses of ONE scan): start poses, results and the scan in pinned, device-mapped host
  // memory -- the kernel reads each start pose once and writes each result once, straight over PCIe, no copy command either way
  void* h_hyp_pinned = nullptr;
  size_t h_hyp_cap = 0;
  // single-scan fast path: endpoints staged in pinned, device-mapped host memory and read by the
  // matcher ONCE (they stay in VGPRs); results written by the kernel straight into h_small
  float2* h_scan_pinned = nullptr;
  size_t h_scan_pinned_cap = 0;
  // ingested scan (hsm_ingest_laser_scan): device container + host copy, sensor trig table cache
  float* d_ranges = nullptr;
  void* d_trig = nullptr;           // float2 (running-angle table) or double2 (laser_geometry unit vectors)
  int trig_kind = -1;
  float ingest_origo[2] = {0.f, 0.f};
  float2* d_ingest = nullptr;
  size_t ingest_cap = 0;
  std::vector<float> h_ingest;      // endpoints as the matcher/updater see them (host copy)
  int ingest_n = -1;                // -1 = nothing ingested yet
  float trig_a0 = 0.f, trig_inc = 0.f;
  int trig_n = -1;
  signed char* d_occ = nullptr;     // occupancy export staging
  size_t d_occ_cap = 0;
  unsigned coop_bar_base = 0;   // value the grid-barrier counter has when the next cooperative launch starts
  float* d_partials = nullptr;  // [2][64][9] per-workgroup partial sums of gn_match_coop_kernel
  int coop_min_beams = 4096;    // single scans at least this long take the multi-workgroup matcher (env HSM_COOP_MIN)
  bool coop_tagged = true;      // env HSM_COOP_TAGGED=0: the counter grid barrier instead of the tagged-record exchange
  void* d_cells = nullptr;  // interleaved {logodds, updateIndex} staging for hsm_download_cells
  size_t d_cells_cap = 0;
  int bpl_override = -1;  // 0 = force the memory loop (env HSM_BPL=0), -1 = auto
  int exact_batch_form = 2;      // env HSM_EXACT_BATCH: 0 = the one-wavefront-per-scan exact form for batches, too; 1 = producer / chain workgroups on maps <= 2^23 cells only (the rule until the <8,2> shape); 2 = on every map
  int xcd_chunk_exact = 0;       // env HSM_XCD_CHUNK_EXACT: the same for the exact-order texel-cache form (0 = contiguous eighths, its default)
  int xcd_chunk = 16;            // env HSM_XCD_CHUNK: workgroups per chunk of the chunked-cyclic batch mapping (0 = contiguous eighths)
  unsigned long long* clock_probe = nullptr;  // hsm_set_clock_probe
  bool cached_wps2 = false;      // env HSM_CACHED_WPS2=1: with waves_per_scan = 2, batches use the two-wave texel-cache form (experimental)
  int spb_large = 8;             // env HSM_SPB_LARGE=4|8: scans per workgroup of the texel-cache matcher on maps > 2^23 cells
  int wg_sync = -1;              // env HSM_WG_SYNC=0|1: per-beam workgroup barrier of the texel-cache matcher (-1 = for maps > 2^23 cells)
  int exact_shape = 0;           // env HSM_EXACT_SHAPE=7|8: producers per workgroup of the exact batch form (0 = by batch size)
  bool dense_bits = true;        // env HSM_DENSE_BITS=0: dense scans keep the keyed update (map_update.h)
  bool exact_cached = true;      // env HSM_EXACT_CACHED=0: exact-mode batches keep round 2's producer / chain-wavefront form (gn_match.h)
  bool exact = false;     // HSM_PARITY_EXACT: H / dTr summed in the reference's beam order (gn_match.h exact_round)
  bool auto_parity = true;  // HSM_PARITY_AUTO (default): every entry point in the reference's summation order (auto_wants_exact)
  bool relaxed = false;   // HSM_PARITY_RELAXED: contracted multiply-adds in the throughput kernel (gn_match_cached_kernel<.., RELAXED>)
  int last_cfg[6] = {0, 0, 0, 0, 0, 0};
  int coop_mute_block = 0;      // hsm_debug_set_coop_mute (test hook)
  bool exact_spec = false;       // env HSM_EXACT_SPEC=1: one-workgroup-per-scan launches in exact order take the speculative-carry form (gn_match_spec.h:
                                 // the same bits; measured SLOWER than the literal chains on one CU -- DESIGN.md 8 -- hence opt-in)
  bool exact_spec1 = false;      // env HSM_EXACT_SPEC1=1: ONE scan of up to 2048 beams (hsm_match) in exact order takes the on-chip speculative-carry form
  // hsm_set_batch_order: launch order of a batch (texel-cache batch forms).  One permutation buffer per stream that has launched
  // a sorted batch (launches on one stream are ordered; a ninth stream keeps the caller's order)
  int batch_order = 2;           // HSM_ORDER_AUTO
  int batch_order_min = 1024;    // env HSM_BATCH_ORDER_MIN: smaller batches keep the caller's order
  int batch_order_refresh = 16;  // hsm_set_batch_order_refresh / env HSM_BATCH_ORDER_REFRESH: a stream's permutation serves that many
                                 // launches of the same batch size before it is computed again (ANY permutation gives the same
                                 // results; an old one only groups the scans by where they were)
  struct PermBuf {
    hipStream_t s;
    int* d;
    size_t cap;
    int batch;  // the batch size the permutation in `d` was computed for (0: none)
    int used;   // launches it has served
  };
  std::vector<PermBuf> perm_bufs;
  bool last_sorted = false;
  float* d_spec_scratch = nullptr;   // gn_match_spec_kernel: products of every beam, [batch][stride] float4s
  size_t spec_scratch_cap = 0;       // float4s
  SpecStats* d_spec_stats = nullptr; // hsm_debug_spec_stats
  bool exact_dense = true;       // env HSM_EXACT_DENSE=0: dense scans in exact order keep the 16-wavefront team form (gn_match_kernel<16,...,EXACT>)
  int exact_dense_min = 4096;
  int compute_units = 256;   // of this device (hsm_create)
  int exact_split_tail = 1;  // env HSM_EXACT_SPLIT_TAIL=0: one launch however the batch divides into generations
  int exact_chain_wave = 1;  // env HSM_EXACT_CHAIN_WAVE=0: no chain-only wavefront, teams of wavefronts for batches below 4096 scans (rounds 3-4)    // env HSM_EXACT_DENSE_MIN: beams from which the producers-ahead-of-the-chain form takes over
  const char* last_kernel = "";  // name of the matcher kernel the last launch used (hsm_last_launch_kernel)
  unsigned coop_fallbacks = 0;  // dense single-scan matches re-run on one workgroup after an exchange timeout (match_single)
  // ... and the back-off that follows: after a timeout the multi-workgroup form is skipped for the next coop_skip matches (1, 2, 4, ...
  // up to 1024, reset by the first exchange that completes) -- on a device another process keeps busy every dense match would
  // otherwise pay the full bounded wait, a host spin and a second launch (round-5 advisor)
  unsigned coop_skip = 0, coop_backoff = 0;
  bool fused_exchange_done = false;  // the last launch_match carried MatchParams::xp itself (else the caller launches the exchange step)
  int last_parity = HSM_PARITY_FAST;  // the mode the last match launch actually ran in (hsm_last_launch_parity)
};

namespace hsm_host {

// match_exact_cached.hip: reference order, batches, one wavefront per scan with the texel cache (by scan length and by how many
// workgroups the launch leaves a CU)
int launch_match_exact_cached_forms(hsm_ctx* h, const hsm::MatchParams& P, int max_n, hipStream_t stream);
// match_teams.hip: `wps` wavefronts per scan (1, 2, 4, 8, 16), either summation order; the one-wavefront exact form goes on to
// launch_match_exact_cached_forms where that applies
// MatchParams::perm for this launch where hsm_set_batch_order asks for it (a sort kernel on `stream` in front of the matcher)
int ensure_batch_perm(hsm_ctx* h, hsm::MatchParams& P, hipStream_t stream);
int launch_match_by_width(hsm_ctx* h, const hsm::MatchParams& P, int max_n, hipStream_t stream, bool exact, int wps);

}  // namespace hsm_host
