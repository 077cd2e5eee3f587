/* hector_mi355 -- C ABI of the MI355X-native scan-to-map Gauss-Newton matcher.
 *
 * This is the drop-in boundary for hector_mapping's L1 map representation
 * (SURVEY.md section 8(b)).  Every entry point names the reference interface it
 * replaces; paths are relative to
 *   /root/reference/hector_mapping/include/hector_slam_lib/        (HSL/)
 * The header facade include/hector_slam_lib/slam_main/MapRepMultiMap.h forwards
 * the reference's C++ virtuals to these functions.
 *
 * Conventions (identical to the reference):
 *   poses   float[3] = x, y, theta; world frame = metres / rad
 *   scans   a DataContainer (HSL/scan/DataPointContainer.h:36-96) is the triple
 *           (pts, n, origo): n endpoints as AoS float[2*n] in the robot frame,
 *           already multiplied by the level-0 scaleToMap (cell units), plus the
 *           laser origin `origo` in the same units
 *   cov/H   float[9], column-major 3x3 (Eigen::Matrix3f default layout)
 *   planes  row-major, index = y*sizeX + x (HSL/map/GridMapBase.h:141-144)
 *
 * Return value: 0 = HSM_OK, negative = error (hsm_last_error() has the text).
 * The reference itself has no error channel (SURVEY.md 8(b) "error conventions");
 * the facade ignores the status to stay behaviour-compatible.
 *
 * All compute runs on the GPU.  There is NO CPU fallback: without a HIP device
 * hsm_create fails with HSM_ERR_NO_DEVICE.
 */
#ifndef HECTOR_MI355_CAPI_H
#define HECTOR_MI355_CAPI_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hsm_ctx hsm_ctx;

enum {
  HSM_OK = 0,
  HSM_ERR_INVALID = -1,    /* bad argument */
  HSM_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime error at init */
  HSM_ERR_HIP = -3,        /* HIP runtime error (text in hsm_last_error) */
  HSM_ERR_TOO_LARGE = -4   /* scan longer than HSM_MAX_UPDATE_BEAMS in update_by_scan, or map > 2^28 cells */
};

#define HSM_MAX_LEVELS 8
#define HSM_MAX_UPDATE_BEAMS 1048575

/* probability sampling layout used by the GN kernel (DESIGN.md "data layout"):
 *   QUAD  float4 texel plane {P(x,y),P(x+1,y),P(x,y+1),P(x+1,y+1)}: one 16-byte gather per beam; best for batched matching
 *   PLANE the fp32 probability plane itself, four 4-byte gathers per beam; no texel plane is kept (16 B/cell less
 *         memory, one dense pass less per update): best for update-heavy single dense scans */
enum { HSM_LAYOUT_AUTO = 0, HSM_LAYOUT_QUAD = 1, HSM_LAYOUT_PLANE = 2 };

typedef struct hsm_opts {
  int device;          /* HIP device ordinal; -1 = current device */
  int layout;          /* HSM_LAYOUT_*; AUTO honours env HSM_LAYOUT=quad|plane, default quad */
  int waves_per_scan;  /* 0 = auto (env HSM_WPS overrides), else 1,2,4,8,16 */
} hsm_opts;

/* summation order of the Hessian / gradient reduction (OccGridMapUtil.h:76-98):
 *   FAST   lane-strided partial sums + tree: the reference's per-beam products, summed in another order;
 *          poses within 1e-4 m / 1e-4 rad of the reference wherever its Gauss-Newton iteration has settled
 *   EXACT  the reference's order, beam 0 .. n-1 in nine sequential fp32 chains: H, dTr, every GN step and
 *          the final pose are BIT-IDENTICAL to the reference CPU matcher on every scan (about 2.5x the
 *          instructions per GN iteration).  sinf/cosf/expf are glibc's algorithms in every mode.
 *   RELAXED  FAST with the multiply-add pairs of the per-beam arithmetic contracted to fused operations (32 instead of
 *          51 fp32 operations per beam) in the batched throughput kernel; everything else runs as FAST.  Per-beam terms
 *          are no longer bit-exact; the bar is north_star's 1e-4 m / 1e-4 rad on the pose, measured at full size.
 *   AUTO   (default) the reference's order on EVERY entry point -- hsm_match (with and without the hook trace), hsm_match_level,
 *          hsm_match_batch*, hsm_group_match_batch*, the likelihood / covariance / Hessian probes: results bit-identical to the
 *          reference CPU matcher.  AUTO may pick any kernel form that is bit-identical to the reference's chains (today the
 *          same forms EXACT pins).  Why the default does not use the tree anywhere (rounds 4 and 5): the scene sweeps
 *          (profiles/r04/parity_scene_sweep.jsonl, profiles/r05/parity_scene_sweep_single_default.jsonl: seven scene families,
 *          three start / level set-ups, 4096 scans each) find the fast tree beyond 1e-4 m of the reference on some scans of every
 *          family wherever the reference's own iteration has not settled, and nothing known at launch separates those scans --
 *          for one scan no more than for a batch.  Price: a 1081-beam hsm_match takes 99 us instead of 33, a batch 1.3x the
 *          tree's time.  hsm_last_launch_parity() tells which order the last launch ran in.
 * env HSM_PARITY=fast|exact|relaxed|auto selects a mode at hsm_create (any other word: hsm_create fails), hsm_set_parity
 * switches at run time. */
enum { HSM_PARITY_FAST = 0, HSM_PARITY_EXACT = 1, HSM_PARITY_RELAXED = 2, HSM_PARITY_AUTO = 3 };

/* ---- construction ---------------------------------------------------------
 * replaces: MapRepMultiMap::MapRepMultiMap(mapResolution, mapSizeX, mapSizeY, numDepth,
 *           startCoords, draw, debug)                    HSL/slam_main/MapRepMultiMap.h:48-72
 * Level l has size (sx>>l, sy>>l) and cell length res*2^l; all levels share the
 * offset (res*sx*start_x, res*sy*start_y).  Update factors start at the library
 * defaults 0.4 / 0.6 (HSL/map/GridMapLogOdds.h:117-118). */
int hsm_create(float map_resolution, int size_x, int size_y, unsigned levels, float start_x,
               float start_y, const hsm_opts* opts /* may be NULL */, hsm_ctx** out);
/* replaces: MapRepMultiMap::~MapRepMultiMap                   MapRepMultiMap.h:74-81
 * Waits for the context's queued work, releases everything.  Never fails towards the caller; if a runtime call fails during
 * teardown, everything else is still released, hsm_last_error() of this thread then reads "hsm_destroy: <call>: <error>", and
 * the runtime's per-thread error state is cleared (it is not left for the next hipGetLastError() of an unrelated call). */
void hsm_destroy(hsm_ctx* h);

/* replaces: MapRepMultiMap::reset -> MapProcContainer::reset  MapRepMultiMap.h:83-90, MapProcContainer.h:67-71 */
int hsm_reset(hsm_ctx* h);
/* replaces: getMapLevels / getScaleToMap                      MapRepMultiMap.h:92-94 */
int hsm_levels(const hsm_ctx* h);
float hsm_scale_to_map(const hsm_ctx* h);
/* replaces: setUpdateFactorFree / setUpdateFactorOccupied     MapRepMultiMap.h:149-167 */
int hsm_set_update_factor_free(hsm_ctx* h, float free_factor);
int hsm_set_update_factor_occupied(hsm_ctx* h, float occupied_factor);
/* replaces: onMapUpdated (cache generation bump)              MapRepMultiMap.h:107-114.
 * The device probability texels are refreshed eagerly by update_by_scan, so this
 * only has to exist; it returns HSM_OK. */
int hsm_on_map_updated(hsm_ctx* h);
/* no reference counterpart: selects HSM_PARITY_FAST / _EXACT / _RELAXED / _AUTO for all later matches of the context */
int hsm_set_parity(hsm_ctx* h, int mode);
int hsm_parity(const hsm_ctx* h);
/* the order the LAST match launch of this context actually ran in (HSM_PARITY_FAST / EXACT / RELAXED); under HSM_PARITY_AUTO:
 * EXACT */
int hsm_last_launch_parity(const hsm_ctx* h);
/* no reference counterpart (the reference matches one scan at a time): the order in which a BATCH is laid out on the device.
 * Scans that are neighbours in a launch run on one XCD at the same time and share the texel lines they touch in its L2; a batch
 * whose order does not follow the map (particles of several modes, several robots) loses that -- 61 instead of 58 us on the
 * 2048^2 map, 178 instead of 145 us on the 4096^2 pyramid (profiles/r06).
 *   HSM_ORDER_MORTON: a one-workgroup counting sort in front of the matcher orders batches of at least 1024 scans by the Morton
 *     code of the 64 x 64 map tile their start poses lie in, and the matcher takes its scans through that permutation; every
 *     result still lands at the scan's own index and is bit-identical.  The sort is a 9 us kernel, so a stream's permutation
 *     serves hsm_set_batch_order_refresh launches of the same batch size (default 16; env HSM_BATCH_ORDER_REFRESH) before it is
 *     computed again: ANY permutation gives the same results, an old one only groups the scans by where they were a few
 *     launches ago.
 *   HSM_ORDER_AUTO (default; env HSM_BATCH_ORDER=auto|given|morton): that, on maps of more than 2^23 cells (the ones that outgrow
 *     the L2s), and only for a batch that does not follow the map already -- the sort kernel counts the tile changes between
 *     neighbouring scans and leaves a batch that has few of them in its own order.  Costs such a batch ~1 us per launch.
 *   HSM_ORDER_GIVEN: always the caller's order.
 * Applies to the texel-cache batch forms (every batch of the quad layout). */
enum { HSM_ORDER_GIVEN = 0, HSM_ORDER_MORTON = 1, HSM_ORDER_AUTO = 2 };
int hsm_set_batch_order(hsm_ctx* h, int order);
int hsm_set_batch_order_refresh(hsm_ctx* h, int launches);
int hsm_batch_order(const hsm_ctx* h);
/* 1 if the last batched match took its scans through a permutation (Morton order, or the identity AUTO left a batch in), else 0 */
int hsm_last_launch_sorted(const hsm_ctx* h);

/* ---- the hot path -----------------------------------------------------------
 * replaces: MapRepMultiMap::matchData(beginEstimateWorld, dataContainer, covMatrix)
 *           HSL/slam_main/MapRepMultiMap.h:116-132  (-> ScanMatcher::matchData,
 *           HSL/matcher/ScanMatcher.h:54-190, 4 GN steps per coarse level, 6 on level 0).
 * Host pointers.  n == 0: out_pose = begin, cov untouched (ScanMatcher.h:68,189).
 * Like the reference it retains the scan for the coarse levels of the next
 * hsm_update_by_scan (MapRepMultiMap.h:127,143). */
int hsm_match(hsm_ctx* h, const float begin_world[3], const float* pts_xy, int n,
              const float origo[2], float out_pose_world[3], float cov[9]);

/* hsm_match plus a per-GN-step trace for the reference's draw/debug hooks
 * (ScanMatcher.h:100-110: drawArrow(estimate) and addHessianMatrix(H) per loop iteration).
 * trace[12*k .. 12*k+11] = {map-frame estimate after step k [3], H used by step k [9] col-major},
 * steps in schedule order (coarsest level first; 4 per coarse level, 6 on level 0).
 * *steps_written = number of records (0 for an empty scan). */
int hsm_match_trace(hsm_ctx* h, const float begin_world[3], const float* pts_xy, int n,
                    const float origo[2], float out_pose_world[3], float cov[9], float* trace,
                    int trace_cap_steps, int* steps_written);

/* Batched extension (not in the reference): B independent (pose hypothesis, scan)
 * pairs against the read-only pyramid in ONE launch.  DEVICE pointers:
 *   d_begin_world  [B*3]       d_pts_xy [total*2]
 *   d_scan_offsets [B+1] CSR offsets into d_pts_xy in points, or NULL = every
 *                  hypothesis uses the same scan d_pts_xy[0 .. shared_n)
 *   shared_n       with CSR offsets: a sizing HINT (typical beams per scan, 0 = unknown -> 1081); it selects
 *                  the kernel form only -- scans longer than the hint are matched correctly, just slower
 *   d_out_pose     [B*3]       d_out_cov [B*9] or NULL
 * `stream` is a hipStream_t (NULL = default stream); the call is asynchronous.  The library orders it behind
 * every map update queued on the context so far, and the next map update behind it (events; no host wait),
 * per caller stream: matches may be in flight on several caller-owned streams at once.
 * Does not touch the retained-scan state. */
int hsm_match_batch_device(hsm_ctx* h, int batch, const float* d_begin_world,
                           const float* d_pts_xy, const int* d_scan_offsets, int shared_n,
                           float* d_out_pose, float* d_out_cov, void* stream);
/* same with host pointers (copies in, runs, copies out, synchronous) */
int hsm_match_batch(hsm_ctx* h, int batch, const float* begin_world, const float* pts_xy,
                    const int* scan_offsets, int shared_n, float* out_pose, float* out_cov);

/* replaces: MapRepMultiMap::updateByScan(dataContainer, robotPoseWorld)
 *           HSL/slam_main/MapRepMultiMap.h:134-147 -> OccGridMapBase::updateByScan,
 *           HSL/map/OccGridMapBase.h:121-260.  Level 0 uses (pts, n, origo); coarse
 *           levels use the scan retained by the last hsm_match, scaled by 2^-level,
 *           exactly as the reference does.
 *           Returns when the update is QUEUED on the context's stream (pts_xy has been copied and may be
 *           reused at once): every later call on this context -- matches, downloads, further updates --
 *           is ordered behind it, so results are the same as if it had completed; a device fault
 *           surfaces at the next call that waits.  hsm_synchronize() waits explicitly; the environment
 *           variable HSM_ASYNC_UPDATE=0 makes the call itself wait. */
int hsm_update_by_scan(hsm_ctx* h, const float pose_world[3], const float* pts_xy, int n,
                       const float origo[2]);
/* wait until all work queued on the context has completed (and report any device error) */
int hsm_synchronize(hsm_ctx* h);
/* one level, explicit level-scaled container (OccGridMapBase::updateByScan itself) */
int hsm_update_by_scan_level(hsm_ctx* h, int level, const float pose_world[3],
                             const float* pts_level_xy, int n, const float origo_level[2]);

/* ---- single-process multi-GPU group (extension; the reference has no multi-device path) ------------------
 * One replica of the pyramid per listed device (a device may be listed more than once).  Batched matching is
 * sharded contiguously over the replicas, one PERSISTENT host thread per replica (created with the group), each
 * replica on its own stream.  Two forms: host arrays in/out (hsm_group_match_batch: the "gather" of the poses is the
 * D2H copy of each shard into the caller's arrays) and device-resident shards (hsm_group_match_batch_device: the
 * gather of each shard's [n,3] poses to the root replica's device runs over xGMI, no host staging -- as ONE grouped RCCL
 * collective over the group's communicators (ncclCommInitAll on first use; librccl is dlopen'ed then, the library does
 * not link it), or as peer copies; see hsm_group_set_gather).  Map updates are replayed on every replica -- updateByScan
 * is deterministic, so the replicas stay bit-identical.
 * (bench.py's --gpus N uses the other deployment shape: one process per GPU and an RCCL all-gather of device-resident
 * poses through torch.distributed; bench.py --group N drives this one.) */
typedef struct hsm_group hsm_group;
/* how hsm_group_match_batch_device gathers.  AUTO (default; env HSM_GROUP_GATHER=auto|direct|rccl|peer at hsm_group_create):
 * DIRECT -- the device-side exchange below (hsm_exchange_*: every replica's kernel stores its rows into every replica's
 * mailbox, one launch per replica behind its match, no collective, no event) -- whenever the replicas' devices can access
 * each other (or are the same device); else RCCL when librccl loads, every replica sits on its own device and
 * ncclCommInitAll succeeds; else peer copies (hsm_group_gather_note says why).  DIRECT or RCCL asked for explicitly fail
 * instead of falling back. */
enum { HSM_GATHER_AUTO = 0, HSM_GATHER_PEER = 1, HSM_GATHER_RCCL = 2, HSM_GATHER_DIRECT = 3 };
int hsm_group_set_gather(hsm_group* g, int mode);
/* test hook: with the RCCL gather, send EVERY shard -- the root's own too (a send to self) -- through the grouped ncclSend /
 * ncclRecv form that unequal shards take, instead of ncclAllGather.  Lets a one-device box run that form on hardware. */
int hsm_group_debug_force_p2p(hsm_group* g, int on);
/* The partitioning of `total` scans over `world` replicas (contiguous shards, the first total % world hold one more): [begin, end)
 * of shard `rank`.  Used by hsm_group_match_batch and by hector_slam_amd/sharding.py -- one rule for both transports. */
int hsm_shard_bounds(int total, int rank, int world, int* begin, int* end);
/* the mode in effect: HSM_GATHER_DIRECT, HSM_GATHER_RCCL or HSM_GATHER_PEER.  The decision -- for RCCL: dlopen of librccl and
 * ncclCommInitAll over the group's devices, SECONDS on a multi-GPU node -- is taken on first use: by this call, by hsm_group_set_gather(RCCL), or else inside the
 * first hsm_group_match_batch_device.  A latency-sensitive caller asks for the mode once right after hsm_group_create. */
int hsm_group_gather_mode(hsm_group* g);
const char* hsm_group_gather_note(const hsm_group* g);
/* the DIRECT gather, and the RCCL gather with equal shards, are all-gathers: replica i (other than the root, which received into
 * the caller's arrays) holds all poses [sum(counts) * 3] (want_cov: all Hessians [sum * 9]) in a block the group owns; NULL otherwise */
const float* hsm_group_gathered(hsm_group* g, int replica, int want_cov);
int hsm_group_create(float map_resolution, int size_x, int size_y, unsigned levels, float start_x, float start_y,
                     const int* devices, int n_devices, hsm_group** out);
void hsm_group_destroy(hsm_group* g);
int hsm_group_size(const hsm_group* g);
hsm_ctx* hsm_group_member(hsm_group* g, int i); /* replica i: uploads, downloads, queries */
int hsm_group_set_update_factors(hsm_group* g, float free_factor, float occupied_factor);
/* HectorSlamProcessor::update's two halves for ONE scan: matchData on replica 0, then -- if do_update --
 * updateByScan with the matched pose on EVERY replica (each retains the scan first, like its own matchData would) */
int hsm_group_process_scan(hsm_group* g, const float hint_world[3], const float* pts_xy, int n, const float origo[2],
                           int do_update, float out_pose_world[3], float cov[9]);
/* Device-resident sharded match: replica r matches counts[r] scans that already live on ITS device (d_begin_world[r],
 * d_pts_xy[r], d_scan_offsets[r]: CSR offsets relative to the shard, or d_scan_offsets == NULL / entries NULL for
 * pose hypotheses of one shared scan of shared_n beams per replica).  The poses of all shards are gathered, in replica
 * order, into d_out_pose_all [sum(counts) * 3] on replica `root`'s device (and the Hessians into d_out_cov_all
 * [sum * 9] unless NULL): by the device-side exchange (one launch per replica on its own stream), by a grouped ncclAllGather
 * (equal counts) / ncclSend + ncclRecv (ragged counts) on the replicas' own streams, or by hipMemcpyPeerAsync on each replica's
 * own stream (hsm_group_set_gather).  Asynchronous: returns when
 * everything is queued; root's context stream is ordered behind the gather (hsm_synchronize(hsm_group_member(g, root)) or
 * hsm_group_synchronize wait for it). */
int hsm_group_match_batch_device(hsm_group* g, const int* counts, const float* const* d_begin_world,
                                 const float* const* d_pts_xy, const int* const* d_scan_offsets, int shared_n, int root,
                                 float* d_out_pose_all, float* d_out_cov_all);
/* wait for all queued work of every replica */
int hsm_group_synchronize(hsm_group* g);
/* hsm_match_batch over all replicas: scans [B*r/R, B*(r+1)/R) go to replica r */
int hsm_group_match_batch(hsm_group* g, int batch, const float* begin_world, const float* pts_xy,
                          const int* scan_offsets, int shared_n, float* out_pose, float* out_cov);
/* what hsm_match does to the coarse-level containers, without matching (MapRepMultiMap.h:127) */
int hsm_retain_scan(hsm_ctx* h, const float* pts_xy, int n, const float origo[2]);

/* ---- device-side gather of sharded results (extension; the reference has no multi-device path) -------------------
 * The one exchange step of a batched matchData sharded over G GPUs (MapRepMultiMap.h:116-132 is the call being sharded;
 * SURVEY.md 8(e)): every rank ends up with every rank's [B/G, cols] rows (cols = 3 poses, 9 Hessians).  No collective library
 * on the data path: every rank owns a MAILBOX in its own HBM; a POST stores a rank's rows -- one 8-byte {value, epoch tag}
 * store per float, system scope -- into every rank's mailbox over xGMI (its own included); a WAIT polls the own mailbox until
 * every value of that epoch has arrived and unpacks it into a dense fp32 array.  One small kernel per step on the caller's
 * stream (protocol and flow control: hector_slam_amd/csrc/pose_exchange.h).  Used by hsm_group_* (HSM_GATHER_DIRECT)
 * and, one process per GPU, by hector_slam_amd/sharding.py (DirectRowGather: the handles travel once over torch.distributed,
 * the rows never do).
 *
 *   create          rank `rank` of `world` (<= 16) on `device` (-1 = current): a mailbox of `depth` buffers of
 *                   [total_rows][cols] values.  depth >= 2 + 2 * lag (lag: how many posts a rank runs ahead of its waits).
 *   handle          64-byte hipIpcMemHandle of this rank's mailbox -- hand it to the other PROCESSES (any channel)
 *   connect         handles of all ranks, world x 64 bytes in rank order (the own entry is ignored): maps the peers' mailboxes
 *   connect_local   the ranks of ONE process: ranks[r] = rank r's exchange (peer access instead of IPC)
 *   post            epoch = posted + 1: rows [first_row, first_row + n_rows) of the gathered array, from d_rows [n_rows][cols]
 *                   (device memory, written by work queued earlier on `stream`), to every rank
 *   wait            epoch = waited + 1 (must have been posted by this rank): d_out_all [total_rows][cols] holds all ranks'
 *                   rows when the launch completes (NULL: arrival only).  Bounded: values that do not arrive within
 *                   HSM_EXCHANGE_TIMEOUT_MS (default 2000) read NaN and hsm_exchange_status fails.
 *   post_wait       ONE launch: post epoch e = posted + 1 and wait for epoch e - lag (skipped while e <= lag, or when that epoch
 *                   has been waited for already)
 *   status          HSM_OK, or HSM_ERR_HIP after a timed-out wait (text in hsm_last_error)
 * Every rank calls post the same number of times; post and wait of one exchange go on ONE stream (or streams the caller orders). */
typedef struct hsm_exchange hsm_exchange;
#define HSM_EXCHANGE_HANDLE_BYTES 64
#define HSM_EXCHANGE_MAX_WORLD 16
int hsm_exchange_create(int device, int rank, int world, int total_rows, int cols, int depth, hsm_exchange** out);
void hsm_exchange_destroy(hsm_exchange* x);
int hsm_exchange_handle(hsm_exchange* x, void* handle64);
int hsm_exchange_connect(hsm_exchange* x, const void* handles);
int hsm_exchange_connect_local(hsm_exchange* x, hsm_exchange* const* ranks);
int hsm_exchange_post(hsm_exchange* x, const float* d_rows, int first_row, int n_rows, void* stream);
int hsm_exchange_wait(hsm_exchange* x, float* d_out_all, void* stream);
int hsm_exchange_post_wait(hsm_exchange* x, const float* d_rows, int first_row, int n_rows, int lag, float* d_out_all,
                           void* stream);
/* hsm_match_batch_device + one exchange step of its poses (post this rank's `batch` rows at `first_row`, wait for the epoch `lag`
 * matches back into d_out_all or NULL) as ONE call: where the matcher form can, the launch carries the exchange itself -- every
 * wavefront posts its pose from the kernel's epilogue and a few extra workgroups at the end of the grid unpack -- so nothing at all
 * runs between two matcher launches; otherwise the stand-alone exchange kernel is queued behind the matcher.  Same epochs, same
 * mailbox contents either way (hsm_exchange_wait drains what is still in flight). */
int hsm_match_batch_device_gather(hsm_ctx* h, int batch, const float* d_begin_world, const float* d_pts_xy,
                                  const int* d_scan_offsets, int shared_n, float* d_out_pose, float* d_out_cov,
                                  hsm_exchange* x, int first_row, int lag, float* d_out_all, void* stream);
int hsm_exchange_epochs(const hsm_exchange* x, unsigned long long* posted, unsigned long long* waited);
int hsm_exchange_status(hsm_exchange* x);
/* "uncached" or "fine-grained": the kind of device memory the mailbox got */
const char* hsm_exchange_memory_kind(const hsm_exchange* x);

/* ---- host mirror support: replaces getGridMap(level) cell access --------------
 * HSL/slam_main/MapRepMultiMap.h:95, HSL/map/GridMapBase.h:141-159 */
int hsm_level_info(const hsm_ctx* h, int level, int* size_x, int* size_y, float* cell_length,
                   float* scale_to_map);
/* GridMapBase::getMapCoordsPose / getWorldCoordsPose         GridMapBase.h:226-239 */
int hsm_map_coords_pose(const hsm_ctx* h, int level, const float world[3], float map[3]);
int hsm_world_coords_pose(const hsm_ctx* h, int level, const float map[3], float world[3]);
/* GridMapBase::getUpdateIndex (bumped by every update, GridMapBase.h:343-344) */
int hsm_update_index(const hsm_ctx* h, int level);
/* whole level; either pointer may be NULL */
int hsm_download_level(hsm_ctx* h, int level, float* logodds, int* update_index);
int hsm_upload_level(hsm_ctx* h, int level, const float* logodds, const int* update_index);
/* rows [y0, y1) of the log-odds plane only (cheap mirror refresh after an update) */
int hsm_download_rows(hsm_ctx* h, int level, int y0, int y1, float* logodds_rows);
/* cells [x0..x1] x [y0..y1] (inclusive) as the reference's LogOddsCell AoS
 * {float logOddsVal; int updateIndex} (GridMapLogOdds.h:89-100), written to dst_cells (= address
 * of cell (x0,y0) in a host grid whose rows are dst_pitch_cells cells apart): the host-mirror
 * refresh behind getGridMap() */
int hsm_download_cells(hsm_ctx* h, int level, int x0, int y0, int x1, int y1, void* dst_cells,
                       int dst_pitch_cells);
/* cell bounding box {x0, y0, x1, y1} (inclusive) touched by the last update of `level`;
 * x1 < x0 when nothing was touched */
int hsm_last_update_bbox(const hsm_ctx* h, int level, int bbox[4]);
/* union of the cell boxes touched on `level` since the previous call (reset/upload mark the whole level),
 * then cleared: what a host mirror has to re-download.  x1 < x0 when nothing changed. */
int hsm_take_dirty_bbox(hsm_ctx* h, int level, int bbox[4]);

/* ---- the rows either side of the path; reference = the ROS node,
 *      hector_mapping/src/HectorMappingRos.cpp.  Optional: the facade does not need them. ---- */
/* replaces: rosLaserScanToDataContainer (HectorMappingRos.cpp:483-507): raw LaserScan ranges ->
 * DataContainer endpoints (fp32 running angle, range gate (range_min, range_max - 0.1f), ordered
 * compaction), computed on the device; the container STAYS on the device as the "ingested scan"
 * (origo = 0,0 like the reference).  The cos/sin table of the sensor geometry (angle_min,
 * angle_increment, n) is evaluated once on the host with the float libm calls the node uses and cached.
 * out_pts_xy (host, capacity 2*n floats, may be NULL) receives the endpoints, *out_n their count. */
int hsm_ingest_laser_scan(hsm_ctx* h, const float* ranges, int n, float angle_min, float angle_increment,
                          float range_min, float range_max, float scale_to_map, float* out_pts_xy, int* out_n);
/* replaces: rosPointCloudToDataContainer (HectorMappingRos.cpp:509-542), the node's DEFAULT ingestion
 * (use_tf_scan_transformation = true, :82,:257-282): n geometry_msgs::Point32 {x,y,z} floats in the laser
 * frame + the laser->base tf::Transform as 12 doubles, rows [R | t] (tfScalar is double).  Gates as the node:
 * x*x+y*y in (sqr_laser_min_dist, sqr_laser_max_dist), x < 0 && dist_sqr < 0.5 dropped, base-frame z minus
 * t_z in (laser_z_min, laser_z_max); endpoint = float(base x,y) * scale_to_map, ordered compaction.  The
 * container stays on the device with origo = float(t_x, t_y) * scale_to_map (also written to out_origo,
 * may be NULL); out_pts_xy capacity 2*n floats, may be NULL. */
int hsm_ingest_point_cloud(hsm_ctx* h, const float* pts_xyz, int n, const double tf_rows[12],
                           float sqr_laser_min_dist, float sqr_laser_max_dist, float laser_z_min, float laser_z_max,
                           float scale_to_map, float* out_pts_xy, int* out_n, float out_origo[2]);
/* the same with the step before it fused in: laser_geometry::LaserProjection::projectLaser(scan, cloud,
 * range_cutoff) (HectorMappingRos.cpp:273; third-party package, not in the reference tree -- restated from
 * its published algorithm: point = float((double)range * (cos, sin)(angle_min + (double)i * angle_increment)),
 * kept when range < range_cutoff && range >= range_min; range_cutoff < 0 means range_max), so raw
 * LaserScan.ranges[] (4 B/beam) is the wire format on the tf path as well. */
int hsm_ingest_laser_scan_tf(hsm_ctx* h, const float* ranges, int n, float angle_min, float angle_increment,
                             float range_min, float range_max, double range_cutoff, const double tf_rows[12],
                             float sqr_laser_min_dist, float sqr_laser_max_dist, float laser_z_min,
                             float laser_z_max, float scale_to_map, float* out_pts_xy, int* out_n,
                             float out_origo[2]);
/* hsm_match / hsm_update_by_scan on the ingested scan (no endpoint upload; origo as ingested) */
int hsm_match_ingested(hsm_ctx* h, const float begin_world[3], float out_pose_world[3], float cov[9]);
int hsm_update_by_ingested(hsm_ctx* h, const float pose_world[3]);
/* replaces: OccGridMapUtil::getLikelihoodForState (HSL/map/OccGridMapUtil.h:184-214) evaluated for
 * `batch` MAP-frame states of `level` against one scan (pts: level-0 units, scaled by 2^-level here):
 * out_lh[b] = 1 - sum_i(1 - M_i)/n -- the particle-weight primitive of the batched-hypotheses use case.
 * Host pointers. */
int hsm_likelihood_states(hsm_ctx* h, int level, int batch, const float* states_map, const float* pts_xy, int n,
                          float* out_lh);
/* replaces: OccGridMapUtil::getResidualForState (HSL/map/OccGridMapUtil.h:205-221): out_residual[b] =
 * sum_i (1 - M_i), same arguments as hsm_likelihood_states. */
int hsm_residual_states(hsm_ctx* h, int level, int batch, const float* states_map, const float* pts_xy, int n,
                        float* out_residual);
/* replaces: OccGridMapUtil::getCovarianceForPose (HSL/map/OccGridMapUtil.h:106-160) and
 * getCovMatrixWorldCoords (:162-188) for `batch` MAP-frame poses of `level`: seven sigma points per pose
 * (x +- 1.5 cells, y +- 1.5 cells, angle +- 0.05 rad, the pose), likelihood-weighted sample covariance.
 * out_cov_map / out_cov_world: batch x 9 floats, column major; out_lh7: batch x 7 likelihoods in the
 * reference's sigma-point order.  Any of the three may be NULL.  (The reference prints the likelihoods to
 * stdout on every call, :136; this entry does not.) */
int hsm_covariance_for_poses(hsm_ctx* h, int level, int batch, const float* poses_map, const float* pts_xy, int n,
                             float* out_cov_map, float* out_cov_world, float* out_lh7);
/* replaces: hectormaptools::DistanceMeasurementProvider::getDist (hector_map_tools/include/hector_map_tools/
 * HectorMapTools.h:133-234, behind hector_map_server's services) for `n` rays on `level`, with the
 * OccupancyGrid metadata the node publishes (origin = getWorldCoords(0,0) - cell/2, resolution = cell length,
 * HectorMappingRos.cpp:546-553): out_dist[i] = resolution * (int) cell distance to the first occupied cell, or
 * -resolution when there is none; out_hit_xy[2i..] = its world coordinates (left untouched without a hit).
 * Host pointers; out_hit_xy may be NULL. */
int hsm_ray_distances(hsm_ctx* h, int level, float origin_x, float origin_y, float resolution, int n,
                      const float* begin_world_xy, const float* end_world_xy, float* out_dist, float* out_hit_xy);
/* replaces: publishMap's cell loop (HectorMappingRos.cpp:449-468) with LogOddsCell::isFree/isOccupied
 * (GridMapLogOdds.h:76-84): -1 unknown, 0 free (logOdds < 0), 100 occupied (logOdds > 0).
 * out: host, sx*sy bytes, row major. */
int hsm_occupancy_grid(hsm_ctx* h, int level, signed char* out);

/* ---- parity / debug entry points (used by tests, not by the facade) ------------ */
/* device probability plane p = e^l/(e^l+1) (GridMapLogOdds.h:163-166) */
int hsm_download_prob(hsm_ctx* h, int level, float* prob);
/* one evaluation of OccGridMapUtil::getCompleteHessianDerivs (OccGridMapUtil.h:64-104)
 * at a MAP-frame pose with level-scaled points */
int hsm_hessian_derivs(hsm_ctx* h, int level, const float pose_map[3], const float* pts_level_xy,
                       int n, float H[9], float dTr[3]);
/* per-beam terms of the same evaluation: out[4*i] = M, dM/dx, dM/dy, rotDeriv */
int hsm_eval_beams(hsm_ctx* h, int level, const float pose_map[3], const float* pts_level_xy,
                   int n, float* out4);
/* ScanMatcher::matchData on one level with explicit max_iterations (ScanMatcher.h:54) */
int hsm_match_level(hsm_ctx* h, int level, const float begin_world[3], const float* pts_level_xy,
                    int n, int max_iterations, float out_pose_world[3], float cov[9]);

/* test hook: set the 12-bit per-scan generation counter of `level`'s key planes (it wraps every 4095
 * updates -- 27 minutes at 40 Hz -- and the wrap path has to be exercised without running that long) */
int hsm_debug_set_update_serial(hsm_ctx* h, int level, unsigned serial);
/* test hook: out[0] = non-zero 32-bit words of `level`'s crossed-cell byte map (dense scans), out[1] = non-zero words of its
 * end-cell bitmap (scans below 4096 beams).  Both planes carry no generation tag: every update's apply pass clears exactly
 * what its mark passes set, so BETWEEN updates both counts must be 0 (waits for the queued updates first) */
int hsm_debug_marks_nonzero(hsm_ctx* h, int level, unsigned long long out[2]);
/* test hook: set the monotonic arrival counter of the cooperative matcher's grid barrier (it advances by
 * workgroups x GN steps per dense match and wraps at 2^32) */
int hsm_debug_set_coop_barrier(hsm_ctx* h, unsigned value);
/* gn_match_spec_kernel (one scan per workgroup, reference order, speculative carries: csrc/spec_chain.h): counters of its
 * stitching passes since the last call -- out[0] segments settled, [1] unused (0), [2] accepted by the shift rule, [3] re-run
 * literally.  enable != 0 keeps counting (zeroed by every call), 0 stops.  Results never
 * depend on these numbers: they only say how fast the exact form ran. */
int hsm_debug_spec_stats(hsm_ctx* h, int enable, unsigned long long out[4]);
/* test hook: workgroup `block_plus_one - 1` of the multi-workgroup dense matcher (HSM_PARITY_FAST / _RELAXED, >= 4096 beams)
 * never publishes its partial sums (0 = off) -- the exchange of every workgroup then times out, which is what a device that
 * cannot keep the K workgroups co-resident looks like.  hsm_match handles that by matching the scan again on the
 * one-workgroup matcher; hsm_debug_coop_fallbacks() counts how often it had to. */
int hsm_debug_set_coop_mute(hsm_ctx* h, int block_plus_one);
int hsm_debug_coop_fallbacks(hsm_ctx* h);
/* device sincosf of n angles (glibc's algorithm, csrc/libm_exact.h) -- numerics test hook */
int hsm_debug_sincos(hsm_ctx* h, int n, const float* x, float* s, float* c);
/* device expf(x) and getGridProbability(x) = e/(e+1) of n values -- numerics test hook */
int hsm_debug_expf(hsm_ctx* h, int n, const float* x, float* out_exp, float* out_prob);

/* the context's device: {HIP ordinal, compute units, shader clock kHz, memory clock kHz} (roofline arithmetic of bench.py) */
int hsm_device_info(const hsm_ctx* h, int info[4]);

/* Measurement aid (no reference counterpart): d_stamps4 = device pointer to four 64-bit words, or NULL to switch it
 * off.  While set, a batched match by a texel-cache form (tree summation: the wavefront of scan 0; reference-order
 * summation: the first wavefront of workgroup 0 of the 4096-scan form, which launches a probed instantiation that is 0.4 us
 * slower) stores {shader-clock counter (s_memtime), 100 MHz wall clock} as taken at its start [0,1] and at its end [2,3]:
 * (s[2]-s[0]) / (s[3]-s[1]) * 100 MHz = the clock the kernel actually ran at (bench.py prices the VALU roof at it next
 * to the nominal 2.4 GHz).  The caller owns the memory. */
int hsm_set_clock_probe(hsm_ctx* h, unsigned long long* d_stamps4);
/* GN steps one full hsm_match performs per scan (4 per coarse level + 6) */
int hsm_gn_iterations_per_match(const hsm_ctx* h);
/* effective kernel configuration of the last match launch:
 * {layout, waves_per_scan, block, grid, beams_per_lane (0 = endpoints streamed from memory; negative =
 * the texel-cache form: that many beams per lane with endpoints in LDS and the last texel of every beam
 * kept in VGPRs)} */
int hsm_last_launch_config(const hsm_ctx* h, int cfg[5]);
/* name of the matcher kernel of the last match launch ("gn_match_exact_cached_kernel", "gn_match_exact_dense_kernel", ...): what
 * bench.py looks up in the rocprofv3 kernel trace; a static string */
const char* hsm_last_launch_kernel(const hsm_ctx* h);

const char* hsm_last_error(void);
const char* hsm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HECTOR_MI355_CAPI_H */
