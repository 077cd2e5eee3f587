// Issue rate of ONE wavefront running a dependent fp32 add chain on gfx950 (tools only):
//   hipcc --offload-arch=gfx950 -O3 -o ubench_chain tools/ubench_chain.hip && ./ubench_chain
// One wavefront per SIMD (256-thread workgroups, one per CU, pinned by a 100 KB LDS allocation).  Shader cycles per v_add_f32:
//   chain      : v_add_f32 v0, v1, v0 back to back (every add waits for the one before)
//   chain/32   : the same with exec = lanes 0..31 only    (does a half-empty wavefront issue in one pass?)
//   chain/16   : ... lanes 0..15
//   2 chains   : two independent chains interleaved
//   4 chains   : four
//   lds        : the chain job's pattern -- ds_read_b128 + s_waitcnt + 4 dependent adds
//   pk chain   : v_pk_add_f32 v[0:1], v[2:3], v[0:1] back to back (round 5: two chains per lane?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters, float b) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = b;
    lds[threadIdx.x + 256] = b;
    __syncthreads();
    float r0 = threadIdx.x, r1 = 1.0f, r2 = 2.0f, r3 = 3.0f;
    unsigned long long saved = 0;
    if (MODE == 1) asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffffffff" : "=s"(saved));
    if (MODE == 2) asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff" : "=s"(saved));
    const unsigned addr = (threadIdx.x & 63) * 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE <= 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r0) : "v"(b));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %2, %0\n\tv_add_f32 %1, %2, %1" : "+v"(r0), "+v"(r1) : "v"(b));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_add_f32 %0, %4, %0\n\tv_add_f32 %1, %4, %1\n\tv_add_f32 %2, %4, %2\n\tv_add_f32 %3, %4, %3"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(b));
        } else if (MODE == 6) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 r = {r0, r1}, bb = {b, b};
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r) : "v"(bb));
            r0 = r.x, r1 = r.y;
        } else {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 a, c;
            asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(addr));
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(c) : "v"(addr));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("s_waitcnt lgkmcnt(1)\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0"
                             : "+v"(r0) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
                asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(addr));
                asm volatile("s_waitcnt lgkmcnt(1)\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0"
                             : "+v"(r0) : "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w));
                asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(c) : "v"(addr));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(c));
            r1 += a.x + c.x;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (MODE == 1 || MODE == 2) asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* d_out, unsigned long long* d_cyc) {
    const int iters = 2000, blocks = 256;
    size_t lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int r = 0; r < 5; ++r) probe<MODE><<<blocks, 256, lds>>>(d_out, d_cyc, iters, 0.5f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= h.size();
    printf("{\"mode\": \"%s\", \"cycles_per_v_add\": %.3f}\n", name, mean / (iters * 32.0));
}

int main() {
    float* d_out;
    unsigned long long* d_cyc;
    (void)hipMalloc(&d_out, 256 * 256 * 4);
    (void)hipMalloc(&d_cyc, 256 * 4 * 8);
    run<0>("chain", d_out, d_cyc);
    run<1>("chain, exec = lanes 0..31", d_out, d_cyc);
    run<2>("chain, exec = lanes 0..15", d_out, d_cyc);
    run<3>("2 chains interleaved", d_out, d_cyc);
    run<4>("4 chains interleaved", d_out, d_cyc);
    run<5>("ds_read_b128 + wait + 4 adds (32 adds per iteration)", d_out, d_cyc);
    run<6>("v_pk_add_f32 chain (cycles per packed add)", d_out, d_cyc);
    return 0;
}
