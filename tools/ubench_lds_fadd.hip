// What the LDS does with fp32 atomic adds whose lanes collide on one address (gfx950; tools only):
//   hipcc --offload-arch=gfx950 -O3 -o ubench_lds_fadd tools/ubench_lds_fadd.hip && ./ubench_lds_fadd
// Question behind it: the reference sums H / dTr beam by beam (OccGridMapUtil.h:76-98); on the VALU that is a chain of
// dependent v_add_f32 at 8.5 cycles each (tools/ubench_chain.hip).  If ONE ds_add_f32 whose 64 lanes name the same word
// performed its 64 additions in ascending lane order, rounding each like v_add_f32, the LDS itself would be the chain.
//   order   : per wavefront, 64 values of widely varying magnitude (so that every order rounds differently) are added to one
//             word with ds_add_rtn_f32; the final word and every lane's returned old value are compared with the sequential
//             fp32 sums in ascending and descending lane order (CPU, same rounding, denormals kept)
//   order9  : lane j adds to word j % 9 (7-8 lanes per word, nine words in nine banks): ascending order within each word?
//   rate    : shader cycles per ds_add_f32 instruction, 64 lanes on one word / nine words / 64 distinct words (the same
//             words again and again: every instruction depends on the one before through memory)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

__global__ __launch_bounds__(64) void order_probe(const float* __restrict__ vals, const float* __restrict__ init, float* __restrict__ fin,
                                                  float* __restrict__ olds, int nine) {
    __shared__ float w[16];
    const int lane = threadIdx.x, trial = blockIdx.x;
    if (lane < 16) w[lane] = init[trial];
    __syncthreads();
    const float v = vals[trial * 64 + lane];
    const unsigned addr = (unsigned)(size_t)&w[nine ? lane % 9 : 0];
    float old;
    asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(v) : "memory");
    __syncthreads();
    olds[trial * 64 + lane] = old;
    if (lane < 9) fin[trial * 9 + lane] = w[lane];
}

template <int MODE>  // 0: one word, 1: nine words, 2: 64 distinct words
__global__ __launch_bounds__(64) void rate_probe(float* out, unsigned long long* cyc, int iters) {
    __shared__ float w[64];
    const int lane = threadIdx.x;
    w[lane] = 0.0f;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)&w[MODE == 0 ? 0 : (MODE == 1 ? lane % 9 : lane)];
    const float v = 1.0f + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(v) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[blockIdx.x * 64 + lane] = w[lane];
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

// the exact-order matcher's pattern: every wavefront of a workgroup adds 64-lane rows to its OWN nine words (word 9 w + t of
// the workgroup: neighbouring banks), `per_iter` rows per loop iteration; WAVES wavefronts per workgroup, `blocks_per_cu`
// workgroups per CU (the launch has 256 * blocks_per_cu workgroups).  Cycles per row of nine instructions, per wavefront.
template <int WAVES, int TERMS>
__global__ __launch_bounds__(64 * WAVES) void load_probe(float* out, unsigned long long* cyc, int iters, int pad_words) {
    extern __shared__ float w[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < WAVES * 16; i += 64 * WAVES) w[i] = 0.0f;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)&w[wave * TERMS];
    const float v = 1.0f + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < TERMS; ++t) asm volatile("ds_add_f32 %0, %1 offset:%2" : : "v"(base), "v"(v), "n"(4 * t) : "memory");
        if ((it & 1) == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x < WAVES * TERMS) out[blockIdx.x * WAVES * TERMS + threadIdx.x] = w[threadIdx.x];
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
}

static float seq_sum(float s, const float* v, const int* idx, int n, float* olds) {
    for (int k = 0; k < n; ++k) {
        if (olds) olds[idx[k]] = s;
        volatile float t = s + v[idx[k]];  // one rounding per addition
        s = t;
    }
    return s;
}

int main() {
    const int trials = 8192;
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> mant(1.0f, 2.0f);
    std::uniform_int_distribution<int> expo(-12, 12), sign(0, 1), rare(0, 31);
    std::vector<float> vals(trials * 64), init(trials);
    for (int t = 0; t < trials; ++t) {
        init[t] = std::ldexp(mant(rng), expo(rng)) * (sign(rng) ? 1.0f : -1.0f);
        for (int l = 0; l < 64; ++l) {
            float x = std::ldexp(mant(rng), expo(rng)) * (sign(rng) ? 1.0f : -1.0f);
            if (rare(rng) == 0) x = std::ldexp(mant(rng), -140);  // a denormal now and then
            if (rare(rng) == 1) x = 0.0f;
            vals[t * 64 + l] = x;
        }
    }
    // a few trials that live among the denormals entirely
    for (int t = 0; t < 64; ++t) {
        init[t] = std::ldexp(mant(rng), -135);
        for (int l = 0; l < 64; ++l) vals[t * 64 + l] = std::ldexp(mant(rng), -130 - (l % 15)) * (sign(rng) ? 1.0f : -1.0f);
    }
    float *d_vals, *d_init, *d_fin, *d_olds;
    (void)hipMalloc(&d_vals, vals.size() * 4);
    (void)hipMalloc(&d_init, init.size() * 4);
    (void)hipMalloc(&d_fin, trials * 9 * 4);
    (void)hipMalloc(&d_olds, vals.size() * 4);
    (void)hipMemcpy(d_vals, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_init, init.data(), init.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> fin(trials * 9), olds(trials * 64);
    for (int nine = 0; nine < 2; ++nine) {
        order_probe<<<trials, 64>>>(d_vals, d_init, d_fin, d_olds, nine);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(fin.data(), d_fin, fin.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(olds.data(), d_olds, olds.size() * 4, hipMemcpyDeviceToHost);
        long asc_fin = 0, desc_fin = 0, asc_olds = 0, desc_olds = 0, words = 0, lanes = 0, order_sensitive = 0;
        for (int t = 0; t < trials; ++t) {
            const int nw = nine ? 9 : 1;
            for (int wd = 0; wd < nw; ++wd) {
                int idx[64], n = 0;
                for (int l = 0; l < 64; ++l)
                    if (!nine || l % 9 == wd) idx[n++] = l;
                int rev[64];
                for (int k = 0; k < n; ++k) rev[k] = idx[n - 1 - k];
                float oa[64], od[64];
                const float fa = seq_sum(init[t], &vals[t * 64], idx, n, oa), fd = seq_sum(init[t], &vals[t * 64], rev, n, od);
                const float got = fin[t * 9 + wd];
                ++words;
                order_sensitive += std::memcmp(&fa, &fd, 4) != 0;
                asc_fin += std::memcmp(&got, &fa, 4) == 0;
                desc_fin += std::memcmp(&got, &fd, 4) == 0;
                for (int k = 0; k < n; ++k) {
                    const int l = idx[k];
                    ++lanes;
                    asc_olds += std::memcmp(&olds[t * 64 + l], &oa[l], 4) == 0;
                    desc_olds += std::memcmp(&olds[t * 64 + l], &od[l], 4) == 0;
                }
            }
        }
        printf("{\"test\": \"%s\", \"words\": %ld, \"order_sensitive_words\": %ld, \"final_equals_ascending\": %ld, \"final_equals_descending\": %ld, "
               "\"lanes\": %ld, \"returned_old_equals_ascending_prefix\": %ld, \"returned_old_equals_descending_prefix\": %ld}\n",
               nine ? "order9" : "order", words, order_sensitive, asc_fin, desc_fin, lanes, asc_olds, desc_olds);
    }
    // where does the first disagreement with the ascending order sit (one-word test)?  print one trial's returned values' ranks
    {
        order_probe<<<trials, 64>>>(d_vals, d_init, d_fin, d_olds, 0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(olds.data(), d_olds, olds.size() * 4, hipMemcpyDeviceToHost);
        // rank lanes of trial 100 by matching the returned old value against running sums: greedy reconstruction of the order
        const int t = 100;
        std::vector<int> order;
        std::vector<bool> used(64, false);
        float s = init[t];
        for (int k = 0; k < 64; ++k) {
            int hit = -1;
            for (int l = 0; l < 64; ++l)
                if (!used[l] && std::memcmp(&olds[t * 64 + l], &s, 4) == 0) {
                    hit = l;
                    break;
                }
            if (hit < 0) break;
            used[hit] = true;
            order.push_back(hit);
            volatile float tmp = s + vals[t * 64 + hit];
            s = tmp;
        }
        printf("{\"test\": \"reconstructed_order_trial_100\", \"lanes_placed\": %zu, \"order\": [", order.size());
        for (size_t k = 0; k < order.size(); ++k) printf("%s%d", k ? ", " : "", order[k]);
        printf("]}\n");
    }
    float* d_out;
    unsigned long long* d_cyc;
    (void)hipMalloc(&d_out, 256 * 64 * 4);
    (void)hipMalloc(&d_cyc, 256 * 8);
    const int iters = 2000;
    auto rate = [&](auto kernel, const char* name) {
        for (int r = 0; r < 3; ++r) kernel<<<256, 64>>>(d_out, d_cyc, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        printf("{\"test\": \"rate\", \"lanes_to_words\": \"%s\", \"cycles_per_ds_add_f32\": %.2f}\n", name, mean / h.size() / (iters * 8.0));
    };
    rate(rate_probe<0>, "64 lanes -> 1 word");
    rate(rate_probe<1>, "64 lanes -> 9 words");
    rate(rate_probe<2>, "64 lanes -> 64 words");
    // the matcher's pattern under load
    float* d_o2;
    unsigned long long* d_c2;
    (void)hipMalloc(&d_o2, 4096 * 16 * 16 * 4);
    (void)hipMalloc(&d_c2, 4096 * 16 * 8);
    auto load = [&](auto kernel, int waves, int terms, int blocks_per_cu, size_t lds_bytes, const char* name) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        const int blocks = 256 * blocks_per_cu, it = 1000;
        for (int r = 0; r < 3; ++r) kernel<<<blocks, 64 * waves, lds_bytes>>>(d_o2, d_c2, it, 0);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)blocks * waves);
        (void)hipMemcpy(h.data(), d_c2, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= h.size();
        printf("{\"test\": \"load\", \"shape\": \"%s\", \"wavefronts_per_cu\": %d, \"terms\": %d, \"cycles_per_row_of_%d_adds_per_wavefront\": %.1f, "
               "\"cycles_per_ds_add_f32_per_cu\": %.2f}\n", name, waves * blocks_per_cu, terms, terms, mean / it, mean / it / (terms * waves * blocks_per_cu));
    };
    load(load_probe<1, 9>, 1, 9, 1, 40 * 1024, "1 wavefront per CU, 9 words");
    load(load_probe<1, 1>, 1, 1, 1, 40 * 1024, "1 wavefront per CU, 1 word");
    load(load_probe<4, 9>, 4, 9, 1, 40 * 1024, "4 wavefronts (1 workgroup) per CU");
    load(load_probe<4, 9>, 4, 9, 4, 40 * 1024, "16 wavefronts (4 workgroups) per CU");
    load(load_probe<16, 9>, 16, 9, 1, 40 * 1024, "16 wavefronts (1 workgroup) per CU");
    return 0;
}
