// The block-scan form of the reference's fp32 chains (tools/study/exact_scan.h) against the literal dependent chain,
// one wavefront each, on REAL per-beam products (tools/study/binade_stats.py --dump: chain-major fp32, nine chains of one
// Gauss-Newton step), staged in LDS as the matcher stages them.  Per chain: both sums (must be the same bits, and the bits of
// the host's literal loop), shader cycles of each form, scan iterations and single additions.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I tools/study -o tools/_bin/ubench_exact_scan tools/study/ubench_exact_scan.hip
//   ./ubench_exact_scan /tmp/products_16384_0_converged.bin 15398
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "exact_scan.h"

constexpr int E = 7, BLK = 64 * E;
constexpr int kMaxLds = 36 * 1024;  // floats of one chain staged at a time (144 KB)

// mode 0: literal chain (lane 0, 16-byte LDS reads, dependent v_add_f32); mode 1: serial prologue of 64 + block scans
template <int MODE>
__global__ __launch_bounds__(64) void chain_kernel(const float* __restrict__ prod, int n, float* out, long long* cyc, int* counters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int npad = (n - 64 + BLK - 1) / BLK * BLK + 64;  // first 64 serial, then whole blocks (zero padded)
  const float* x = prod + (size_t)blockIdx.x * n;
  for (int i = lane; i < npad; i += 64) lds[i] = i < n ? x[i] : 0.0f;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  float s = 0.0f;
  int iters = 0, singles = 0;
  if (MODE == 0) {
    if (lane == 0) {
      const float4* row = reinterpret_cast<const float4*>(lds);
      for (int q = 0; q < npad / 4; ++q) {
        const float4 v = row[q];
        asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0" : "+v"(s) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
      }
    }
    s = __shfl(s, 0);
  } else {
    if (lane == 0) {
      const float4* row = reinterpret_cast<const float4*>(lds);
      for (int q = 0; q < 16; ++q) {
        const float4 v = row[q];
        asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0" : "+v"(s) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
      }
    }
    s = __shfl(s, 0);
    for (int b = 64; b < npad; b += BLK) {
      float xv[E];
#pragma unroll
      for (int j = 0; j < E; ++j) xv[j] = lds[b + lane * E + j];
      s = hsm::xscan::block_sum<E>(s, xv, 0, lane, &iters, &singles);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) {
    out[blockIdx.x] = s;
    cyc[blockIdx.x] = t1 - t0;
    counters[2 * blockIdx.x] = iters;
    counters[2 * blockIdx.x + 1] = singles;
  }
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s products.bin n [chains=9]\n", argv[0]); return 2; }
  const int n = atoi(argv[2]), chains = argc > 3 ? atoi(argv[3]) : 9;
  std::vector<float> h((size_t)n * chains);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(h.data(), 4, h.size(), f) != h.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  fclose(f);
  if (n + BLK > kMaxLds) { fprintf(stderr, "n too large for one LDS stage\n"); return 2; }
  float *d_prod, *d_out;
  long long* d_cyc;
  int* d_cnt;
  (void)hipMalloc(&d_prod, h.size() * 4);
  (void)hipMalloc(&d_out, chains * 4);
  (void)hipMalloc(&d_cyc, chains * 8);
  (void)hipMalloc(&d_cnt, chains * 8);
  (void)hipMemcpy(d_prod, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const size_t lds = (size_t)(n + BLK + 64) * 4;
  (void)hipFuncSetAttribute((const void*)chain_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)chain_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<float> o0(chains), o1(chains);
  std::vector<long long> c0(chains), c1(chains);
  std::vector<int> cnt(2 * chains);
  for (int rep = 0; rep < 3; ++rep) {
    chain_kernel<0><<<chains, 64, lds>>>(d_prod, n, d_out, d_cyc, d_cnt);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(o0.data(), d_out, chains * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c0.data(), d_cyc, chains * 8, hipMemcpyDeviceToHost);
    chain_kernel<1><<<chains, 64, lds>>>(d_prod, n, d_out, d_cyc, d_cnt);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 3; }
    (void)hipMemcpy(o1.data(), d_out, chains * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c1.data(), d_cyc, chains * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(cnt.data(), d_cnt, chains * 8, hipMemcpyDeviceToHost);
  }
  int bad = 0;
  for (int c = 0; c < chains; ++c) {
    volatile float s = 0.0f;
    for (int i = 0; i < n; ++i) s = s + h[(size_t)c * n + i];
    const float want = s;
    unsigned a, b, w;
    memcpy(&a, &o0[c], 4); memcpy(&b, &o1[c], 4); memcpy(&w, &want, 4);
    const bool ok = a == w && b == w;
    bad += !ok;
    printf("{\"chain\": %d, \"n\": %d, \"literal_cycles\": %lld, \"scan_cycles\": %lld, \"speedup\": %.2f, \"scan_iterations\": %d, \"single_adds\": %d, "
           "\"blocks\": %d, \"cycles_per_iteration\": %.0f, \"bits\": \"%08x %08x %08x\", \"ok\": %s}\n",
           c, n, c0[c], c1[c], (double)c0[c] / (double)c1[c], cnt[2 * c], cnt[2 * c + 1], (n - 64 + BLK - 1) / BLK,
           (double)(c1[c] - 64 * 9) / (cnt[2 * c] > 0 ? cnt[2 * c] : 1), w, a, b, ok ? "true" : "false");
  }
  return bad ? 1 : 0;
}
