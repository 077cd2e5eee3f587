// hsm_host.h -- what the host-side translation units of libhector_mi355.so share: the per-thread error text behind
// hsm_last_error() and the HIP status macro.  (hector_mi355.hip defines hsm_host::fail; the other objects only call it.)
#pragma once
#include <hip/hip_runtime.h>

namespace hsm_host {

// records "<what>[: <hip error text>]" as this thread's hsm_last_error() and returns `code`
int fail(int code, const char* what, hipError_t e = hipSuccess);

}  // namespace hsm_host

struct hsm_exchange;
namespace hsm {
struct ExchangeFused;
}
namespace hsm_host {
// pose_exchange.hip: the arguments of an exchange step for a matcher launch that carries it (hsm_match_batch_device_gather)
int exchange_fused_begin(hsm_exchange* x, int first_row, int n_rows, int lag, float* d_out_all, hsm::ExchangeFused* out);
void exchange_fused_commit(hsm_exchange* x, const hsm::ExchangeFused& f);
}  // namespace hsm_host

#define HSM_HIP_TRY(expr)                                                     \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess) return ::hsm_host::fail(HSM_ERR_HIP, #expr, e__);   \
  } while (0)
