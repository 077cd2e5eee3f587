// VALU issue-rate probe for gfx950 (tools only, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
// For W = 1..4 wavefronts per SIMD (one workgroup of 4 W wavefronts per CU, pinned by a 100 KB LDS allocation) and a stream of independent fp32 operations per wavefront it prints the
// shader cycles one SIMD spends per wave64 instruction (s_memtime over the wavefront's lifetime / instructions / W):
//   mode 0: v_mul_f32 + v_add_f32           (what the bit-exact matcher is made of)
//   mode 1: v_pk_mul_f32 + v_pk_add_f32     (two fp32 per lane and instruction, same rounding per component)
//   mode 2: v_fma_f32
//   mode 3: v_pk_fma_f32
// The question it answers: does packing two beams into one instruction buy issue slots on this part?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, int iters, float a, float b) {
    extern __shared__ float pad_lds[];  // 100 KB per workgroup: one workgroup per CU, so a block of 4 W wavefronts is W per SIMD
    if (iters < 0) pad_lds[threadIdx.x] = a;
    f2v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f2v{(float)threadIdx.x + i, (float)i - threadIdx.x};
    f2v av = {a, a}, bv = {b, b};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {
                asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(a));
                asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(b));
            } else if (MODE == 4) {  // the same 32 operations, every result first read 16 instructions later
                asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(a));
            } else if (MODE == 1) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(av));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(bv));
            } else if (MODE == 2) {
                asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(a), "v"(b));
            } else {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(av), "v"(bv));
            }
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(b));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, float* d_out, unsigned long long* d_cyc) {
    const int iters = 4000;
    for (int W : {1, 2, 3, 4}) {
        int blocks = 256, threads = 256 * W;
        size_t lds = 100 * 1024;
        (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        const int reps = 30;  // sustained: the engine clock under a full-VALU load settles after a few launches
        for (int r = 0; r < reps; ++r) probe<MODE><<<blocks, threads, lds>>>(d_out, d_cyc, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) probe<MODE><<<blocks, threads, lds>>>(d_out, d_cyc, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        std::vector<unsigned long long> h(blocks * 16);
        (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int bl = 0; bl < blocks; ++bl)
            for (int wv = 0; wv < 4 * W; ++wv) mean += (double)h[bl * 16 + wv];
        mean /= blocks * 4 * W;
        double instr = (double)iters * per_iter;
        printf("{\"mode\": \"%s\", \"waves_per_simd\": %d, \"kernel_us\": %.1f, \"wave_cycles_per_instr\": %.3f, "
               "\"simd_cycles_per_instr\": %.3f, \"simd_ns_per_instr\": %.4f, \"chip_G_instr_per_s\": %.0f, \"shader_clock_GHz\": %.3f}\n",
               name, W, ms * 1e3, mean / instr, mean / instr / W, ms * 1e6 / (instr * W), 1024.0 * instr * W / (ms * 1e6),
               mean / (ms * 1e6));
    }
}

int main() {
    float* d_out;
    unsigned long long* d_cyc;
    (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
    (void)hipMalloc(&d_cyc, 256 * 16 * 8);
    run<0>("mul+add", 32, d_out, d_cyc);
    run<1>("pk_mul+pk_add", 16, d_out, d_cyc);
    run<2>("fma", 16, d_out, d_cyc);
    run<3>("pk_fma", 8, d_out, d_cyc);
    run<4>("mul..add (distance 16)", 32, d_out, d_cyc);
    return 0;
}
