// libm_exact.h -- the three libm calls of the hot path, reproduced BIT FOR BIT.
//
// The reference evaluates, per Gauss-Newton step, sin/cos of the pose angle through the float
// overloads (OccGridMapUtil.h:70-71 -> one sincosf call after GCC's sincos pass) and, per map cell,
// expf of the log-odds value (GridMapLogOdds.h:163-166).  On every x86-64 host with FMA + AVX2 (any
// CPU since 2013, incl. the GPU box's EPYC) glibc >= 2.28 dispatches these to its *_fma ifunc
// variants: the double-precision algorithms of Arm's optimized-routines (sincosf: Cody-Waite style
// reduction by pi/2 through the 2^24-scaled 2/pi product, degree-7/8 minimax polynomials evaluated
// as two fused chains; expf: 32-entry 2^(i/32) table and a cubic), compiled with every a*b+c fused.
// Each operation below is one IEEE-754 binary64 operation with the same operands in the same
// fusion pattern as that build (read off the instruction stream of glibc 2.35's libm.so.6:
// __sincosf_fma / __expf_fma), so the float results are identical for EVERY input -- the sweep in
// tests/test_libm_model.py checks all 2^32 arguments against the host libm on the CPU (this very
// header compiled with g++), and tests/test_gpu_parity.py checks the device against it.
//
// Usable from hipcc (device + host) and from plain g++ (the CPU check).  Build with
// -ffp-contract=off: nothing here may be fused or split by the compiler.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HSM_HD __host__ __device__ __forceinline__
#else
#define HSM_HD static inline
#endif

namespace hsm {
namespace libm {

HSM_HD uint32_t f32_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
HSM_HD float bits_f32(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
HSM_HD uint64_t f64_bits(double f) {
  uint64_t u;
  memcpy(&u, &f, 8);
  return u;
}
HSM_HD double bits_f64(uint64_t u) {
  double f;
  memcpy(&f, &u, 8);
  return f;
}

// A binary64 constant whose materialisation stays next to its use.  On the device the optimiser otherwise hoists the
// ~10 constants of sincosf out of the matcher's Gauss-Newton loop into registers (SGPR pairs, VGPR pairs for second operands) that stay live across the whole beam
// loop -- in kernels that keep a per-beam texel cache in all 128 (gn_match_exact.h: spills, which the asynchronous
// inline-asm gathers cannot tolerate).  The volatile asm is not loop invariant; the value is unchanged.
// PIN = false (the latency forms, which have registers to spare): the plain constant -- hoisting it IS the right thing there
// (re-materialising ten constants costs ~20 scalar moves per Gauss-Newton step on a 14-step dependent chain).
template <bool PIN = true>
HSM_HD double at_use(double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (PIN) asm volatile("" : "+s"(c));
#endif
  return c;
}

// ---- sincosf ----------------------------------------------------------------------------------
// Polynomial stage shared by all argument ranges.  xs = reduced argument times the quadrant sign,
// x2 = (reduced argument)^2, n = quadrant.  Returns the double-precision sine/cosine of the
// reduced argument, already swapped/negated for the quadrant.
//   sin ~ xs + xs^3 S1 + xs^5 (S2 + x2 S3)          cos ~ C0 + x2 C1 + x4 C2 + x6 (C3 + x2 C4)
// Quadrants 2,3 use the table with negated cosine coefficients: every operation of that chain is
// odd in the coefficients, so its result is the exact negation of the first table's.
template <bool PIN = true>
HSM_HD void sincosf_poly(double xs, double x2, int n, float& sinp, float& cosp) {
  const double C0 = 0x1p0, C1 = at_use<PIN>(-0x1.ffffffd0c621cp-2), C2 = at_use<PIN>(0x1.55553e1068f19p-5),
               C3 = at_use<PIN>(-0x1.6c087e89a359dp-10), C4 = at_use<PIN>(0x1.99343027bf8c3p-16);
  const double S1 = at_use<PIN>(-0x1.555545995a603p-3), S2 = at_use<PIN>(0x1.1107605230bc4p-7), S3 = at_use<PIN>(-0x1.994eb3774cf24p-13);
  const double x3 = x2 * xs;
  const double x4 = x2 * x2;
  const double s1 = __builtin_fma(x2, S3, S2);
  const double c2 = __builtin_fma(x2, C4, C3);
  const double x5 = x2 * x3;
  const double x6 = x2 * x4;
  const double c1 = __builtin_fma(x2, C1, C0);
  const double s = __builtin_fma(x3, S1, xs);
  const double c = __builtin_fma(x4, C2, c1);
  const double sd = __builtin_fma(s1, x5, s);
  double cd = __builtin_fma(c2, x6, c);
  if (n & 2) cd = -cd;
  const float sf = (float)sd, cf = (float)cd;
  sinp = (n & 1) ? cf : sf;
  cosp = (n & 1) ? sf : cf;
}

// |y| >= 120: the argument's mantissa times 192 bits of 4/pi (three 32-bit windows of the table
// selected by the exponent), top two bits of the product = quadrant, the rest = signed fraction.
HSM_HD double reduce_large(uint32_t xi, int& np) {
  const uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                                 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                                 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
  const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
  const int shift = (xi >> 23) & 7;
  xi = (xi & 0xffffff) | 0x800000;
  xi <<= shift;
  uint64_t res0 = (uint32_t)(xi * arr[0]);
  const uint64_t res1 = (uint64_t)xi * arr[4];
  const uint64_t res2 = (uint64_t)xi * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  const uint64_t n = (res0 + (1ULL << 61)) >> 62;
  res0 -= n << 62;
  np = (int)n;
  return (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
}

template <bool PIN = true>
HSM_HD void sincosf_glibc(float y, float& sinp, float& cosp) {
  const uint32_t xi = f32_bits(y);
  const uint32_t top = (xi >> 20) & 0x7ff;
  const double x = (double)y;
  if (top < 0x3f4) {  // |y| < pi/4
    if (top < 0x398) {  // |y| < 2^-12: sin = y, cos = 1
      sinp = y;
      cosp = 1.0f;
      return;
    }
    sincosf_poly<PIN>(x, x * x, 0, sinp, cosp);
  } else if (top < 0x42f) {  // |y| < 120
    const double r = x * at_use<PIN>(0x1.45F306DC9C883p+23);  // 2/pi * 2^24
    const int n = ((int32_t)r + 0x800000) >> 24;
    const double xr = __builtin_fma(-(double)n, at_use<PIN>(0x1.921FB54442D18p0), x);
    const double sign = ((n ^ (n >> 1)) & 1) ? -1.0 : 1.0;  // +,-,-,+ for quadrants 0..3
    sincosf_poly<PIN>(xr * sign, xr * xr, n, sinp, cosp);
  } else if (top < 0x7f8) {
    int n;
    const double xr = reduce_large(xi, n);
    const int q = n + (int)(xi >> 31);
    const double sign = ((q ^ (q >> 1)) & 1) ? -1.0 : 1.0;
    // table (cosine sign) from q, swap from n -- as the source does
    sincosf_poly<PIN>(xr * sign, xr * xr, (q & 2) | (n & 1), sinp, cosp);
  } else {  // inf / NaN
    sinp = cosp = y - y;
  }
}

// ---- expf ---------------------------------------------------------------------------------------
// exp(x) = 2^(k/32) * 2^(r/32),  k = round(x * 32/ln2) via the 1.5*2^52 shift, r in [-1/2, 1/2]:
// T[k % 32] carries 2^((k%32)/32) with the exponent contribution of k folded in by integer add.
HSM_HD float expf_glibc(float x) {
  const uint32_t xi = f32_bits(x);
  const uint32_t top = (xi >> 20) & 0x7ff;
  const double xd = (double)x;
  if (top > 0x42a) {  // |x| >= 88 or NaN
    if (xi == 0xff800000u) return 0.0f;
    if (top > 0x7f7) return x + x;
    if (x > 0x1.62e42ep6f) return bits_f32(0x7f800000u);  // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;                   // underflow
    if (x < -0x1.9d1d9ep6f) return bits_f32(1u);           // may-underflow: 0x1.4p-75f squared rounds to 2^-149
  }
  const uint64_t T[32] = {
      0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
      0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
      0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
      0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
      0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
      0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
      0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
      0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
  const double InvLn2N = 0x1.71547652b82fep+5, Shift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
  double kd = __builtin_fma(InvLn2N, xd, Shift);
  const uint64_t ki = f64_bits(kd);
  kd -= Shift;
  const double r = __builtin_fma(InvLn2N, xd, -kd);
  const double s = bits_f64(T[ki & 31] + (ki << 47));
  const double z = __builtin_fma(r, C0, C1);
  const double r2 = r * r;
  double yv = __builtin_fma(r, C2, 1.0);
  yv = __builtin_fma(z, r2, yv);
  yv = yv * s;
  return (float)yv;
}

}  // namespace libm
}  // namespace hsm
