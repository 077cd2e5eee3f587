// CPU check of hector_slam_amd/csrc/libm_exact.h (the PRODUCT header, compiled here with g++):
// sincosf_glibc / expf_glibc against the host libm's sincosf / expf over every `stride`-th float
// bit pattern (stride 1 = all 2^32 arguments).  Prints one JSON line; exit code 1 on any mismatch.
// Build: g++ -O2 -ffp-contract=off -pthread libm_model_check.cpp -lm
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../hector_slam_amd/csrc/libm_exact.h"

static inline bool same(float a, float b) {
  const uint32_t x = hsm::libm::f32_bits(a), y = hsm::libm::f32_bits(b);
  if (x == y) return true;
  return (a != a) && (b != b);  // any NaN equals any NaN (payload / sign of invalid results not pinned)
}

int main(int argc, char** argv) {
  const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 10) : 257;
  const int T = argc > 2 ? atoi(argv[2]) : (int)std::thread::hardware_concurrency();
  std::atomic<uint64_t> bad_sc{0}, bad_exp{0}, n{0};
  std::atomic<uint32_t> first_sc{0}, first_exp{0};
  std::vector<std::thread> th;
  const uint64_t total = (1ULL << 32);
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      uint64_t lb_sc = 0, lb_exp = 0, ln = 0;
      // interleaved so that every thread sees every exponent range
      for (uint64_t u = (uint64_t)t * stride; u < total; u += stride * (uint64_t)T) {
        const float x = hsm::libm::bits_f32((uint32_t)u);
        float s0, c0, s1, c1;
        sincosf(x, &s0, &c0);
        hsm::libm::sincosf_glibc(x, s1, c1);
        if (!same(s0, s1) || !same(c0, c1)) {
          if (!lb_sc++) first_sc.store((uint32_t)u);
        }
        const float e0 = expf(x), e1 = hsm::libm::expf_glibc(x);
        if (!same(e0, e1)) {
          if (!lb_exp++) first_exp.store((uint32_t)u);
        }
        ++ln;
      }
      bad_sc += lb_sc;
      bad_exp += lb_exp;
      n += ln;
    });
  for (auto& x : th) x.join();
  printf("{\"checked\": %llu, \"stride\": %llu, \"sincosf_mismatches\": %llu, \"expf_mismatches\": %llu, "
         "\"first_bad_sincosf_bits\": \"0x%08x\", \"first_bad_expf_bits\": \"0x%08x\"}\n",
         (unsigned long long)n.load(), (unsigned long long)stride, (unsigned long long)bad_sc.load(),
         (unsigned long long)bad_exp.load(), first_sc.load(), first_exp.load());
  return (bad_sc.load() || bad_exp.load()) ? 1 : 0;
}
