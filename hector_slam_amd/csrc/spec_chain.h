// spec_chain.h -- the reference's sequential fp32 sums, bit for bit, off the critical path: SEGMENTS of a chain run in
// parallel from speculated carries and are stitched together by an exact shift rule, with the literal loop as the fallback.
//
// The chain (OccGridMapUtil::getCompleteHessianDerivs, HSL/map/OccGridMapUtil.h:76-98: nine of them per Gauss-Newton step):
//     s_0 = 0,   s_i = RN(s_{i-1} + x_i)            (one fp32 rounding per beam, beam order)
// costs n dependent additions however many lanes idle (8.5 cycles each on a lone wavefront: 16 384 beams x 14 steps = 0.8 ms).
//
// Shift rule.  Let r = RN(s + x) lie strictly inside a binade whose ulp is u, and let d be a multiple of u such that r + d lies
// inside the same binade.  Then RN((s + d) + x) = r + d, unless s + x is exactly halfway between two floats (a tie rounds to
// even, and an odd d / u flips the parity).  Proof: |s + x - r| <= u/2, so r + d is a nearest point of the u-grid to s + d + x,
// and s + d + x lies in the binade whose grid that is.  By induction a whole segment started from the carry c + d instead of c
// ends at f(c) + d, PROVIDED d is a multiple of the largest ulp any of its results has, every result stays inside its binade
// when shifted, and no step is a tie (or d is a multiple of TWICE the ulp of every tie step).  All three are properties of the
// run from c alone -- a per-segment summary (SegSummary) that costs five independent operations per addition.
//
// Use.  Exact prefix sums (fp64, a parallel scan) give a candidate carry c_j = (float)S_j for every segment boundary; every
// segment runs the LITERAL fp32 loop from its candidate in its own lane and records f_j and the summary; one short sequential
// pass then walks the boundaries: the true carry t_j is known from the segment before, d = t_j - c_j (exact: neighbouring
// floats), and either the summary accepts d -- t_{j+1} = f_j + d, one addition -- or the segment is re-run from t_j, literally.
// Either way t_{j+1} is what the literal loop produces: bit-identical by construction, data only decides the speed.
//
// Shared by the device kernels (gn_match_spec.h), and by the host model tests/cpp/spec_chain_model.cpp that checks the rule
// against the literal loop on recorded and adversarial chains.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define HSM_SHD __host__ __device__ inline
#else
#define HSM_SHD inline
#endif

namespace hsm {
namespace spec {

HSM_SHD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
#endif
}
HSM_SHD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

// What a run of a segment from a candidate carry leaves behind: three running values, five independent operations per
// addition (min, max, the two subtractions of Fast2Sum, max |err|).  The rule is applied in its cheapest sufficient form: ALL
// values of the run -- the carry-in and every result -- must lie in ONE binade with one sign (then one ulp u serves every
// step, |x| <= |s| at every step so Fast2Sum is exact, and a tie shows as max |err| == u / 2); a run that crosses a binade
// only accepts d = 0.  (The general rule -- per-step ulps -- accepts little more on real chains: tools/study/spec_chain_stats.py.)
struct SegSummary {
  float vmin, vmax;  // smallest / largest value of the run, carry-in included
  float emax;        // largest |rounding error| of a step
  HSM_SHD void reset(float carry_in) {
    vmin = carry_in;
    vmax = carry_in;
    emax = 0.0f;
  }
};

// fold one step r = RN(s + x) of the candidate run into the summary.  `s` is the value before the step.
HSM_SHD void seg_step(SegSummary& S, float s, float x, float r) {
  S.vmin = r < S.vmin ? r : S.vmin;
  S.vmax = r > S.vmax ? r : S.vmax;
  const float bb = r - s;    // Fast2Sum (exact when |s| >= |x|: guaranteed inside one binade, irrelevant otherwise)
  const float err = x - bb;
  const float ae = u2f(f2u(err) & 0x7fffffffu);
  S.emax = ae > S.emax ? ae : S.emax;
}

// the admissible shifts of a finished run: lo <= d <= hi, d a multiple of unit; unit == 0: none but d = 0
struct SegShifts {
  float lo, hi, unit;
};
HSM_SHD SegShifts seg_shifts(const SegSummary& S) {
  SegShifts R{0.0f, 0.0f, 0.0f};
  const uint32_t a = f2u(S.vmin), b = f2u(S.vmax);
  if (((a ^ b) & 0xff800000u) != 0u) return R;  // two signs or two binades (NaN: its exponent differs from any finite partner's, or both are NaN: e = 255 below)
  const uint32_t e = (a >> 23) & 0xffu;
  if (e < 26u || e >= 254u) return R;  // zero, denormal, binades whose ulp is not a normal float (|r| < 2^-101), the largest binade, inf, nan
  const float u = u2f((e - 23u) << 23);          // the ulp of the binade
  const float b_lo = u2f(e << 23), b_hi = u2f((e + 1u) << 23);
  const bool neg = (a >> 31) != 0u;
  const float amin = neg ? -S.vmax : S.vmin, amax = neg ? -S.vmin : S.vmax;  // magnitudes
  // every |r| + d' must stay in [2^e + u, 2^(e+1) - 2u]: one grid point of margin on both sides, so that s + d + x -- within u / 2
  // of it -- lies in the binade as well, and a result on the edge (whose own rounding may have used the finer grid below) never shifts
  const float up = (b_hi - amax) - (u + u);  // exact: same binade
  const float dn = (amin - b_lo) - u;
  if (up < 0.0f || dn < 0.0f) return R;
  R.lo = neg ? -up : -dn;
  R.hi = neg ? dn : up;
  R.unit = (S.emax + S.emax == u) ? u + u : u;  // a tie step: the shift must keep the parity of its result
  return R;
}

// d = t - c for a true carry t and the candidate c the segment was run from; exact iff both lie in one binade with one sign
// (then *exact is set; otherwise the caller re-runs the segment)
HSM_SHD float seg_delta(float t, float c, bool* exact) {
  *exact = ((f2u(t) ^ f2u(c)) & 0xff800000u) == 0u;
  return t - c;
}

// may the segment be shifted by d (= true carry - candidate carry)?
HSM_SHD bool seg_accepts(const SegShifts& S, float d) {
  if (d == 0.0f) return true;  // the candidate WAS the carry
  if (!(d >= S.lo && d <= S.hi) || S.unit == 0.0f) return false;
  const float q = d / S.unit;  // a power of two divides exactly; the quotient is an integer iff d is a multiple
  const uint32_t qb = f2u(q) & 0x7fffffffu;
  if (qb >= 0x4b000000u) return qb < 0x7f800000u;  // |q| >= 2^23: integral by construction (finite)
  const float t = (u2f(qb) + 8388608.0f) - 8388608.0f;  // round to integer (|q| < 2^23)
  return t == u2f(qb);
}

// ---- how a chain of n additions is dealt to one wavefront: lane L owns G consecutive segments of m additions each ------------
struct Plan {
  int m;      // additions per segment (a multiple of 4: the products travel as float4)
  int G;      // segments per lane (1 .. 8)
  int lanes;  // lanes that own at least one addition (<= 64)
  int segs;   // segments that hold at least one addition
};
HSM_SHD Plan plan(int n) {
  Plan p;
  int span = (n + 63) / 64;        // additions per lane, before rounding
  span = (span + 3) & ~3;
  if (span < 8) span = 8;
  p.G = (span + 31) / 32;
  if (p.G > 8) p.G = 8;
  p.m = (((span + p.G - 1) / p.G) + 3) & ~3;
  const int per_lane = p.G * p.m;
  p.lanes = (n + per_lane - 1) / per_lane;
  if (p.lanes < 1) p.lanes = 1;
  p.segs = (n + p.m - 1) / p.m;
  return p;
}

}  // namespace spec
}  // namespace hsm
