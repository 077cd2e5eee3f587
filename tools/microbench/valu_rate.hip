// valu_rate.hip -- how many cycles does one wave64 VALU instruction occupy a gfx950 SIMD?
// Measures v_mul_f32 / v_add_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_fma_f32 / v_cvt / v_med3 issue
// cost with 1, 2, 4 and 8 resident waves per SIMD (independent accumulator chains, no memory).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a[8];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.0f}; }
  const float m = 1.0000001f, c = 1e-9f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f2{m, m}));
        if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f2{c, c}));
        if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (OP == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (OP == 6) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
        if (OP == 7) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
        if (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(f2{m, m}));
        if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f2{m, m}), "v"(f2{c, c}));
      }
    }
  }
  long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}

template <int OP>
void run(const char* name, float* d) {
  const int iters = 2000;
  for (int waves_per_simd : {1, 2, 4, 8}) {
    // 256 CUs x 4 SIMDs; block = 256 threads = 4 waves = 1 wave per SIMD of one CU
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, (long long*)d + (1 << 20), 8, hipMemcpyDeviceToHost);
    const double insts = (double)iters * 64;  // per wave
    printf("%-22s waves/SIMD=%d  wall %.3f ms  clock64 ticks/inst/wave %.2f  => wall ns per (inst x waves/SIMD) %.3f\n",
           name, waves_per_simd, ms, cyc / insts, ms * 1e6 / (insts * waves_per_simd));
  }
}

int main() {
  float* d; hipMalloc(&d, (size_t)(1 << 20) * 8 + 64);
  run<0>("v_mul_f32", d); run<1>("v_add_f32", d); run<4>("v_fma_f32", d);
  run<2>("v_pk_mul_f32", d); run<3>("v_pk_add_f32", d); run<8>("v_pk_mul_f32 op_sel", d); run<10>("v_pk_fma_f32", d);
  run<5>("v_med3_f32", d); run<6>("v_fract_f32", d); run<7>("v_cvt_i32_f32", d); run<9>("v_mad_u32_u24", d);
  return 0;
}
