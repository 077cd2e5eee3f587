// pose_exchange.hip -- device-side gather of the [B/G, cols] result rows of a sharded batched match: every rank stores its
// rows straight into every rank's mailbox (its own HBM, IPC- or peer-mapped into the others), no collective library on the data
// path.  Protocol, layout and flow control: pose_exchange.h.  C ABI: hsm_exchange_* (include/hector_mi355/capi.h).
//
// ONE kernel per exchange step, on the stream the matcher runs on: its first workgroups POST (one system-scope 8-byte store per
// value and peer), the rest WAIT (poll the own mailbox with system-scope loads until every granule carries the epoch's tag)
// and unpack.  A wait is bounded: a granule that has not arrived within the timeout (env HSM_EXCHANGE_TIMEOUT_MS, default
// 2000) ends the poll, the row reads NaN and the exchange's status word says so -- a dead peer costs one timeout, not a hung
// device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "hector_mi355/capi.h"
#include "hsm_host.h"
#include "pose_exchange.h"

namespace hsm {

struct ExchangeArgs {
  ExchangeLayout lay;
  uint64_t* peer[kExchangeMaxWorld];  // every rank's mailbox as mapped in this process; peer[rank] is this rank's own
  const float* rows;                  // [n_rows][cols], this rank's results (written by the launch in front of this one)
  float* out;                         // [total_rows][cols] or nullptr (wait: arrival only)
  unsigned* status;                   // pinned host words: [0] granules that timed out, [1] low word of the epoch they belonged to
  unsigned long long post_epoch, wait_epoch;  // 0 = no such part in this launch
  unsigned long long timeout_ticks;           // of the 100 MHz wall clock
  int rank, first_row, n_rows, post_blocks;   // post_blocks per peer
};

constexpr int kExchangeBlock = 256;
constexpr int kExchangeWaitPerThread = 4;

__global__ void __launch_bounds__(kExchangeBlock) pose_exchange_kernel(const ExchangeArgs A) {
  const int n_post = A.post_epoch ? A.post_blocks * A.lay.world : 0;
  const int blk = (int)blockIdx.x;
  if (blk < n_post) {
    // POST: block -> (peer, slice of this rank's granules).  Posting blocks have the lowest indices, so they are dispatched
    // before any block of this launch starts polling.
    const int p = blk / A.post_blocks, b = blk - p * A.post_blocks;
    uint64_t* dst = A.peer[p] + A.lay.buffer_of(A.post_epoch) + (size_t)A.first_row * (size_t)A.lay.cols;
    const int n = A.n_rows * A.lay.cols;
    for (int i = b * kExchangeBlock + (int)threadIdx.x; i < n; i += A.post_blocks * kExchangeBlock)
      __hip_atomic_store(dst + i, exchange_pack(__float_as_uint(A.rows[i]), A.post_epoch), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);  // global_store_dwordx2 ... sc0 sc1
    return;
  }
  if (!A.wait_epoch) return;
  // WAIT + unpack: kExchangeWaitPerThread granules per thread (few polling wavefronts: while they wait they hold slots a peer's
  // -- or, on a shared device, another rank's -- matcher wants)
  const uint64_t* box = A.peer[A.rank] + A.lay.buffer_of(A.wait_epoch);
  const size_t n = A.lay.buffer_granules();
  const size_t base = (size_t)(blk - n_post) * (kExchangeBlock * kExchangeWaitPerThread) + threadIdx.x;
  unsigned long long t0 = 0;
  for (int j = 0; j < kExchangeWaitPerThread; ++j) {
    const size_t i = base + (size_t)j * kExchangeBlock;
    if (i >= n) return;
    uint64_t g = __hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // global_load_dwordx2 ... sc0 sc1
    bool late = false;
    while (!exchange_carries(g, A.wait_epoch)) {
      if (t0 == 0) t0 = wall_clock64() | 1ull;
      __builtin_amdgcn_s_sleep(4);
      g = __hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (!exchange_carries(g, A.wait_epoch) && wall_clock64() - t0 > A.timeout_ticks) {
        late = true;
        break;
      }
    }
    if (late) {  // (t0 stays: the launch as a whole is bounded by one timeout per thread, not one per granule)
      __hip_atomic_fetch_add(A.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(A.status + 1, (unsigned)A.wait_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (A.out) A.out[i] = __uint_as_float(late ? 0x7fc00000u : exchange_value(g));
  }
}

}  // namespace hsm

using namespace hsm;

struct hsm_exchange {
  int device = 0, rank = 0;
  ExchangeLayout lay{};
  uint64_t* mailbox = nullptr;
  uint64_t* peer[kExchangeMaxWorld] = {};
  bool opened[kExchangeMaxWorld] = {};  // peer[r] came from hipIpcOpenMemHandle
  bool connected = false;
  unsigned* status = nullptr;  // hipHostMalloc'ed, mapped
  unsigned long long posted = 0, waited = 0;
  unsigned long long timeout_ticks = 200000000ull;
  const char* memory_kind = "";
  std::mutex mu;
};

namespace {

int exchange_launch(hsm_exchange* x, const float* d_rows, int first_row, int n_rows, unsigned long long post_epoch,
                    unsigned long long wait_epoch, float* d_out_all, void* stream) {
  ExchangeArgs A;
  memset(&A, 0, sizeof A);
  A.lay = x->lay;
  for (int r = 0; r < x->lay.world; ++r) A.peer[r] = x->peer[r];
  A.rows = d_rows;
  A.out = d_out_all;
  A.status = x->status;
  A.post_epoch = post_epoch;
  A.wait_epoch = wait_epoch;
  A.timeout_ticks = x->timeout_ticks;
  A.rank = x->rank;
  A.first_row = first_row;
  A.n_rows = n_rows;
  const int n = n_rows * x->lay.cols;
  A.post_blocks = post_epoch ? (n + kExchangeBlock - 1) / kExchangeBlock : 0;
  if (post_epoch && A.post_blocks < 1) A.post_blocks = 1;
  const size_t per_block = (size_t)kExchangeBlock * kExchangeWaitPerThread;
  const size_t wait_blocks = wait_epoch ? (x->lay.buffer_granules() + per_block - 1) / per_block : 0;
  const size_t grid = (size_t)A.post_blocks * (size_t)x->lay.world + wait_blocks;
  if (grid == 0) return HSM_OK;
  HSM_HIP_TRY(hipSetDevice(x->device));
  hipLaunchKernelGGL(pose_exchange_kernel, dim3((unsigned)grid), dim3(kExchangeBlock), 0, (hipStream_t)stream, A);
  HSM_HIP_TRY(hipGetLastError());
  return HSM_OK;
}

int check_rows(const hsm_exchange* x, const float* d_rows, int first_row, int n_rows, const char* who) {
  if (n_rows < 0 || first_row < 0 || (long long)first_row + n_rows > x->lay.total_rows || (n_rows > 0 && !d_rows))
    return hsm_host::fail(HSM_ERR_INVALID, who);
  return HSM_OK;
}

}  // namespace

extern "C" {

int hsm_exchange_create(int device, int rank, int world, int total_rows, int cols, int depth, hsm_exchange** out) {
  if (!out) return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_create: out is null");
  *out = nullptr;
  if (world < 1 || world > kExchangeMaxWorld || rank < 0 || rank >= world || total_rows < 1 || cols < 1 || depth < 2 ||
      (size_t)total_rows * (size_t)cols > ((size_t)1 << 28))
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_create: bad argument (1 <= world <= 16, depth >= 2)");
  if (device < 0) HSM_HIP_TRY(hipGetDevice(&device));
  HSM_HIP_TRY(hipSetDevice(device));
  hsm_exchange* x = new hsm_exchange();
  x->device = device;
  x->rank = rank;
  x->lay = ExchangeLayout{world, total_rows, cols, depth};
  // Uncached (else fine-grained) device memory: a line of it is never held in an L2 across another agent's store, which is
  // what a mailbox polled while peers write it needs; ordinary hipMalloc memory is only coherent at kernel boundaries.
  hipError_t e = hipExtMallocWithFlags((void**)&x->mailbox, x->lay.bytes(), hipDeviceMallocUncached);
  x->memory_kind = "uncached";
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags((void**)&x->mailbox, x->lay.bytes(), hipDeviceMallocFinegrained);
    x->memory_kind = "fine-grained";
  }
  if (e != hipSuccess) {
    delete x;
    return hsm_host::fail(HSM_ERR_HIP, "hsm_exchange_create: hipExtMallocWithFlags(uncached / fine-grained)", e);
  }
  e = hipMemset(x->mailbox, 0, x->lay.bytes());
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipHostMalloc((void**)&x->status, 2 * sizeof(unsigned), hipHostMallocMapped);
  if (e != hipSuccess) {
    (void)hipFree(x->mailbox);
    delete x;
    return hsm_host::fail(HSM_ERR_HIP, "hsm_exchange_create: mailbox setup", e);
  }
  x->status[0] = x->status[1] = 0;
  x->peer[rank] = x->mailbox;
  if (const char* env = getenv("HSM_EXCHANGE_TIMEOUT_MS")) {
    const long ms = atol(env);
    if (ms > 0) x->timeout_ticks = (unsigned long long)ms * 100000ull;
  }
  x->connected = world == 1;
  *out = x;
  return HSM_OK;
}

void hsm_exchange_destroy(hsm_exchange* x) {
  if (!x) return;
  (void)hipSetDevice(x->device);
  (void)hipDeviceSynchronize();
  for (int r = 0; r < x->lay.world; ++r)
    if (x->opened[r] && x->peer[r]) (void)hipIpcCloseMemHandle(x->peer[r]);
  if (x->mailbox) (void)hipFree(x->mailbox);
  if (x->status) (void)hipHostFree(x->status);
  (void)hipGetLastError();
  delete x;
}

int hsm_exchange_handle(hsm_exchange* x, void* handle64) {
  if (!x || !handle64) return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_handle: null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == HSM_EXCHANGE_HANDLE_BYTES, "capi.h states the size of an IPC handle");
  HSM_HIP_TRY(hipSetDevice(x->device));
  hipIpcMemHandle_t h;
  HSM_HIP_TRY(hipIpcGetMemHandle(&h, x->mailbox));
  memcpy(handle64, &h, sizeof h);
  return HSM_OK;
}

int hsm_exchange_connect(hsm_exchange* x, const void* handles) {
  if (!x || (!handles && x->lay.world > 1)) return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_connect: null argument");
  std::lock_guard<std::mutex> lk(x->mu);
  if (x->connected) return HSM_OK;
  HSM_HIP_TRY(hipSetDevice(x->device));
  for (int r = 0; r < x->lay.world; ++r) {
    if (r == x->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)r * sizeof h, sizeof h);
    void* p = nullptr;
    HSM_HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    x->peer[r] = (uint64_t*)p;
    x->opened[r] = true;
  }
  x->connected = true;
  return HSM_OK;
}

int hsm_exchange_connect_local(hsm_exchange* x, hsm_exchange* const* ranks) {
  if (!x || !ranks) return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_connect_local: null argument");
  std::lock_guard<std::mutex> lk(x->mu);
  HSM_HIP_TRY(hipSetDevice(x->device));
  for (int r = 0; r < x->lay.world; ++r) {
    const hsm_exchange* o = ranks[r];
    if (!o || o->rank != r || o->lay.world != x->lay.world || o->lay.total_rows != x->lay.total_rows || o->lay.cols != x->lay.cols ||
        o->lay.depth != x->lay.depth)
      return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_connect_local: ranks[r] must be rank r of the same exchange shape");
    if (r == x->rank) continue;
    if (o->device != x->device) {
      int can = 0;
      HSM_HIP_TRY(hipDeviceCanAccessPeer(&can, x->device, o->device));
      if (!can) return hsm_host::fail(HSM_ERR_HIP, "hsm_exchange_connect_local: no peer access between the devices");
      const hipError_t e = hipDeviceEnablePeerAccess(o->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return hsm_host::fail(HSM_ERR_HIP, "hipDeviceEnablePeerAccess", e);
      (void)hipGetLastError();
    }
    x->peer[r] = o->mailbox;
  }
  x->connected = true;
  return HSM_OK;
}

int hsm_exchange_post(hsm_exchange* x, const float* d_rows, int first_row, int n_rows, void* stream) {
  if (!x) return hsm_host::fail(HSM_ERR_INVALID, "null exchange");
  std::lock_guard<std::mutex> lk(x->mu);
  if (!x->connected) return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_post: not connected");
  if (int rc = check_rows(x, d_rows, first_row, n_rows, "hsm_exchange_post: rows outside the gathered array")) return rc;
  if (!exchange_post_is_safe(x->posted + 1, x->waited, x->lay.depth))
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_post: too far ahead of this rank's waits for the mailbox depth (depth >= 2 + 2 lag)");
  if (int rc = exchange_launch(x, d_rows, first_row, n_rows, x->posted + 1, 0, nullptr, stream)) return rc;
  ++x->posted;
  return HSM_OK;
}

int hsm_exchange_wait(hsm_exchange* x, float* d_out_all, void* stream) {
  if (!x) return hsm_host::fail(HSM_ERR_INVALID, "null exchange");
  std::lock_guard<std::mutex> lk(x->mu);
  if (!x->connected) return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_wait: not connected");
  if (x->waited >= x->posted) return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_wait: nothing posted that has not been waited for");
  if (int rc = exchange_launch(x, nullptr, 0, 0, 0, x->waited + 1, d_out_all, stream)) return rc;
  ++x->waited;
  return HSM_OK;
}

int hsm_exchange_post_wait(hsm_exchange* x, const float* d_rows, int first_row, int n_rows, int lag, float* d_out_all,
                           void* stream) {
  if (!x) return hsm_host::fail(HSM_ERR_INVALID, "null exchange");
  std::lock_guard<std::mutex> lk(x->mu);
  if (!x->connected) return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_post_wait: not connected");
  if (lag < 0 || exchange_min_depth(lag) > x->lay.depth)
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_post_wait: lag needs a mailbox of depth >= 2 + 2 lag");
  if (int rc = check_rows(x, d_rows, first_row, n_rows, "hsm_exchange_post_wait: rows outside the gathered array")) return rc;
  const unsigned long long e = x->posted + 1;
  unsigned long long w = e > (unsigned long long)lag ? e - (unsigned long long)lag : 0;
  if (w <= x->waited) w = 0;  // already waited for (a drain by hsm_exchange_wait since): this launch only posts
  if (w != 0 && w != x->waited + 1)
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_post_wait: waits must follow each other (mixing lags needs hsm_exchange_wait in between)");
  if (!exchange_post_is_safe(e, w ? w - 1 : x->waited, x->lay.depth))
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_exchange_post_wait: too far ahead of this rank's waits for the mailbox depth");
  if (int rc = exchange_launch(x, d_rows, first_row, n_rows, e, w, d_out_all, stream)) return rc;
  x->posted = e;
  if (w) x->waited = w;
  return HSM_OK;
}

}  // extern "C"

// ---- a matcher launch that carries the exchange itself (gn_match.h: exchange_post_pose / exchange_wait_unpack) ---------------------
// begin: what post_wait would launch, as the arguments a matcher kernel needs (epochs NOT advanced yet); commit: the matcher
// launch that took them has been queued on `stream`.  A matcher form that cannot carry them leaves the step to hsm_exchange_post_wait.
int hsm_host::exchange_fused_begin(hsm_exchange* x, int first_row, int n_rows, int lag, float* d_out_all, hsm::ExchangeFused* out) {
  if (!x || !out) return hsm_host::fail(HSM_ERR_INVALID, "null exchange");
  std::lock_guard<std::mutex> lk(x->mu);
  if (!x->connected) return hsm_host::fail(HSM_ERR_INVALID, "hsm_match_batch_device_gather: exchange not connected");
  if (x->lay.cols != 3) return hsm_host::fail(HSM_ERR_INVALID, "hsm_match_batch_device_gather: the exchange must carry 3-float rows (poses)");
  if (lag < 0 || exchange_min_depth(lag) > x->lay.depth)
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_match_batch_device_gather: lag needs a mailbox of depth >= 2 + 2 lag");
  if (n_rows < 0 || first_row < 0 || (long long)first_row + n_rows > x->lay.total_rows)
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_match_batch_device_gather: rows outside the gathered array");
  const unsigned long long e = x->posted + 1;
  unsigned long long w = e > (unsigned long long)lag ? e - (unsigned long long)lag : 0;
  if (w <= x->waited) w = 0;
  if (w != 0 && w != x->waited + 1)
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_match_batch_device_gather: waits must follow each other");
  if (!exchange_post_is_safe(e, w ? w - 1 : x->waited, x->lay.depth))
    return hsm_host::fail(HSM_ERR_INVALID, "hsm_match_batch_device_gather: too far ahead of this rank's waits for the mailbox depth");
  memset(out, 0, sizeof *out);
  for (int r = 0; r < x->lay.world; ++r) out->peer[r] = x->peer[r];
  out->out = d_out_all;
  out->status = x->status;
  out->post_off = x->lay.buffer_of(e) + (unsigned long long)first_row * 3ull;
  out->wait_off = w ? x->lay.buffer_of(w) : 0;
  out->timeout_ticks = x->timeout_ticks;
  out->post_tag = (unsigned)e;
  out->wait_tag = (unsigned)w;
  out->world = x->lay.world;
  out->rank = x->rank;
  out->cols = 3;
  out->total_granules = (int)x->lay.buffer_granules();
  out->wait_blocks = w ? (int)((x->lay.buffer_granules() + 1023) / 1024) : 0;
  if (out->wait_blocks > 64) out->wait_blocks = 64;
  return HSM_OK;
}

void hsm_host::exchange_fused_commit(hsm_exchange* x, const hsm::ExchangeFused& f) {
  std::lock_guard<std::mutex> lk(x->mu);
  x->posted += 1;
  if (f.wait_blocks > 0) x->waited += 1;
}

extern "C" {

int hsm_exchange_epochs(const hsm_exchange* x, unsigned long long* posted, unsigned long long* waited) {
  if (!x) return hsm_host::fail(HSM_ERR_INVALID, "null exchange");
  if (posted) *posted = x->posted;
  if (waited) *waited = x->waited;
  return HSM_OK;
}

int hsm_exchange_status(hsm_exchange* x) {
  if (!x) return hsm_host::fail(HSM_ERR_INVALID, "null exchange");
  const unsigned late = __atomic_load_n(&x->status[0], __ATOMIC_ACQUIRE);
  if (late == 0) return HSM_OK;
  char b[256];
  snprintf(b, sizeof b, "hsm_exchange: %u values of epoch (low word) %u did not arrive within the timeout (a peer did not post, or posted another row range)",
           late, x->status[1]);
  return hsm_host::fail(HSM_ERR_HIP, b);
}

const char* hsm_exchange_memory_kind(const hsm_exchange* x) { return x ? x->memory_kind : ""; }

}  // extern "C"
