// gn_match.h -- the scan-to-map Gauss-Newton matcher as CDNA4 (gfx950) HIP kernels.
//
// What it computes (reference, HSL/ = hector_mapping/include/hector_slam_lib/):
//   per beam   OccGridMapUtil::interpMapValueWithDerivatives   HSL/map/OccGridMapUtil.h:287-347
//   per scan   OccGridMapUtil::getCompleteHessianDerivs        HSL/map/OccGridMapUtil.h:64-104
//   per step   ScanMatcher::estimateTransformationLogLh        HSL/matcher/ScanMatcher.h:194-221
//   per level  ScanMatcher::matchData                          HSL/matcher/ScanMatcher.h:54-190
//   per match  MapRepMultiMap::matchData                       HSL/slam_main/MapRepMultiMap.h:116-132
//
// How it is mapped onto the machine:
//   * one TEAM of WPS wavefronts (64 lanes each) owns one (pose hypothesis, scan)
//     pair for the WHOLE coarse-to-fine schedule (all levels, all GN steps) -- the
//     14-step dependent chain never leaves the CU, no host round trips;
//   * beams are dealt round-robin to lanes (beam i -> lane i mod 64*WPS), so the
//     float2 endpoint loads of a wavefront are one contiguous 512-byte segment and
//     neighbouring lanes sample neighbouring map cells;
//   * the occupancy pyramid is sampled from a texture-like "quad" plane: texel
//     (x,y) = float4{P(x,y), P(x+1,y), P(x,y+1), P(x+1,y+1)} -> ONE 16-byte gather
//     per beam instead of four 4-byte gathers on two rows (HSM_LAYOUT_PLANE keeps
//     the 4-gather form for A/B measurements);
//   * the 6 unique H terms + 3 dTr terms are lane-local fp32 partial sums, reduced
//     with a wavefront butterfly (__shfl_xor), then -- when WPS > 1 -- staged
//     through LDS (double buffered, one barrier per GN step);
//   * every lane ends up with bit-identical totals and solves the 3x3 system
//     redundantly: no broadcast, no divergence.
//   No MFMA: this is a bilinear gather plus a 9-term reduction, not a contraction.
//
// Numerics: built with -ffp-contract=off.  Every per-beam value (M, dM/dx, dM/dy,
// rotDeriv and the nine products) is the same IEEE fp32 expression, in the same
// order, as the reference; only the ORDER of the beam summation differs (strided
// partial sums + tree instead of one sequential chain).  sin/cos/exp are evaluated
// in fp64 and rounded once to fp32 (within 1 ulp of glibc's sinf/cosf/expf, which
// the reference calls through the float overloads).
#pragma once
#include <hip/hip_runtime.h>

namespace hsm {

constexpr int kMaxLevels = 8;
constexpr int kLayoutQuad = 1;
constexpr int kLayoutPlane = 2;
constexpr int kUnroll = 4;  // texel gathers a lane keeps in flight (register-resident form)

// Eigen::Affine2f as the reference builds it: 2x2 linear (column major) + translation.
struct Affine2 {
  float l00, l10, l01, l11, t0, t1;
};

// read-only view of one pyramid level for the matcher
struct LevelView {
  const float4* quad;  // [sy*sx] texels {P00,P10,P01,P11}
  const float* prob;   // [sy*sx] plain probability plane
  int sx, sy;
  float limx, limy;    // dims - 2  (MapDimensionProperties.h:70-74)
  Affine2 mapTworld;   // GridMapBase.h:272
  Affine2 worldTmap;   // GridMapBase.h:279
  float pt_scale;      // 2^-level applied to the level-0 endpoints (DataPointContainer.h:46-58)
  int gn_steps;        // 1 + maxIterations (ScanMatcher.h:74,94-97)
};

struct MatchParams {
  LevelView lv[kMaxLevels];
  int first_level;         // coarsest level to run (levels first_level .. last_level, descending)
  int last_level;
  int batch;
  const float* begin_world;  // [B*3]
  const float2* pts;         // packed endpoints
  const int* offsets;        // [B+1] or nullptr (shared scan)
  int shared_n;
  float* out_pose;           // [B*3]
  float* out_cov;            // [B*9] or nullptr
  float* trace;              // nullptr, or [steps*12] per-GN-step record of scan 0 (draw/debug hooks):
                             // {map-frame estimate after the step [3], H of that step [9] col-major}
};

// Transform<Affine> * Vector2f = t + (l(i,0)*x + l(i,1)*y)   (Eigen Transform.h)
__device__ __forceinline__ void affine_apply(const Affine2& a, float x, float y, float& ox, float& oy) {
  ox = a.t0 + (a.l00 * x + a.l01 * y);
  oy = a.t1 + (a.l10 * x + a.l11 * y);
}

// sinf/cosf of the reference (float overloads): fp64 then one rounding.
// Lean fp64 kernel instead of the generic libm sincos(double): Cody-Waite reduction by pi/2 in
// two parts (the 33-bit head makes k*head exact for |k| < 2^20) and the classic degree-13/14
// minimax polynomials on [-pi/4, pi/4] (fdlibm k_sin/k_cos coefficients, < 1 ulp in fp64).
// ~25 fp64 instructions, a handful of live registers, no table, no slow path in the hot loop;
// validated to round to the correctly rounded fp32 value (tests sweep it on the device).
__device__ __forceinline__ void sincos_f32(float th, float& s, float& c) {
  double x = (double)th;
  if (!(fabs(x) < 1048576.0)) {
    // unrealistically large angles (the matcher normalises theta after every level):
    // bring them into range first; inf/NaN fall through and yield NaN like libm
    x = fmod(x, 6.283185307179586476925);
  }
  const double k = rint(x * 0.63661977236758134308);
  double r = x - k * 1.57079632673412561417e+00;  // exact product
  r = r - k * 6.07710050650619224932e-11;
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = ps * z + -2.50507602534068634195e-08;
  ps = ps * z + 2.75573137070700676789e-06;
  ps = ps * z + -1.98412698298579493134e-04;
  ps = ps * z + 8.33333333332248946124e-03;
  ps = ps * z + -1.66666666666666324348e-01;
  // x == +-0 keeps its sign (k = +-0 turns x - k*c into +0): sinf(-0.0f) = -0.0f
  const double sr = (x == 0.0) ? x : r + r * z * ps;
  double pc = -1.13596475577881948265e-11;
  pc = pc * z + 2.08757232129817482790e-09;
  pc = pc * z + -2.75573143513906633035e-07;
  pc = pc * z + 2.48015872894767294178e-05;
  pc = pc * z + -1.38888888888741095749e-03;
  pc = pc * z + 4.16666666666666019037e-02;
  const double cr = 1.0 - 0.5 * z + z * z * pc;
  const int q = (int)k & 3;
  const double sd = (q & 1) ? cr : sr;
  const double cd = (q & 1) ? sr : cr;
  s = (float)((q & 2) ? -sd : sd);
  c = (float)(((q + 1) & 2) ? -cd : cd);
}

// util::normalize_angle (HSL/util/UtilFunctions.h:37-49): double fmod, float result
__device__ __forceinline__ float normalize_angle(float angle) {
  const double two_pi = 2.0 * 3.14159265358979323846;
  float a = (float)fmod(fmod((double)angle, two_pi) + two_pi, two_pi);
  if ((double)a > 3.14159265358979323846) {
    a = (float)((double)a - two_pi);
  }
  return a;
}

struct BeamTerms {
  float M, gx, gy;
};

// a1, split in two so that a lane can ISSUE the texel gathers of several beams back to back
// (stage 1) before it CONSUMES any of them (stage 2): memory latency is hidden by
// instruction-level parallelism inside the wavefront, not only by occupancy.
struct BeamSample {
  float i0, i1, i2, i3;  // P(ix,iy), P(ix+1,iy), P(ix,iy+1), P(ix+1,iy+1)
  float fx, fy;
  bool oob;
};

// stage 1: bounds test, cell index, fractions, and the (asynchronous) gather
template <int LAYOUT>
__device__ __forceinline__ BeamSample sample_fetch(const LevelView& L, float cx, float cy, bool live = true) {
  BeamSample b;
  // MapDimensionProperties::pointOutOfMapBounds (MapDimensionProperties.h:65-68); `live` is false
  // for the padding slots of a lane that has fewer beams than its register file holds: they
  // take the same exact-zero path as an out-of-map beam
  b.oob = !live || (cx < 0.0f) || (cx > L.limx) || (cy < 0.0f) || (cy > L.limy);
  // out-of-map lanes sample texel 0 and are zeroed in stage 2 (the reference returns (0,0,0))
  const float sx_ = b.oob ? 0.0f : cx;
  const float sy_ = b.oob ? 0.0f : cy;
  const int ix = (int)sx_;  // truncation, OccGridMapUtil.h:295
  const int iy = (int)sy_;
  b.fx = sx_ - (float)ix;  // :298
  b.fy = sy_ - (float)iy;
  const int index = iy * L.sx + ix;  // :302
  if (LAYOUT == kLayoutQuad) {
    const float4 q = L.quad[index];
    b.i0 = q.x;
    b.i1 = q.y;
    b.i2 = q.z;
    b.i3 = q.w;
  } else {
    // indices index, index+1, index+sizeX, index+sizeX+1 (:306-330).  A NaN coordinate
    // passes the bounds test like in the reference; v_cvt_i32_f32(NaN) = 0 keeps the
    // address inside the plane and the NaN fraction poisons the result as it should.
    const float* p = L.prob + index;
    b.i0 = p[0];
    b.i1 = p[1];
    b.i2 = p[L.sx];
    b.i3 = p[L.sx + 1];
  }
  return b;
}

// stage 2: the interpolation and the source-literal "derivatives" (:332-346)
__device__ __forceinline__ BeamTerms sample_finish(const BeamSample& b) {
  const float dx1 = b.i0 - b.i1;  // :332-336
  const float dx2 = b.i2 - b.i3;
  const float dy1 = b.i0 - b.i2;
  const float dy2 = b.i1 - b.i3;
  const float xFacInv = (1.0f - b.fx);  // :338-339
  const float yFacInv = (1.0f - b.fy);
  BeamTerms r;
  // :341-346, source-literal (x-differences blended with the x fractions)
  r.M = ((b.i0 * xFacInv + b.i1 * b.fx) * (yFacInv)) + ((b.i2 * xFacInv + b.i3 * b.fx) * (b.fy));
  r.gx = -((dx1 * xFacInv) + (dx2 * b.fx));
  r.gy = -((dy1 * yFacInv) + (dy2 * b.fy));
  if (b.oob) {
    r.M = 0.0f;
    r.gx = 0.0f;
    r.gy = 0.0f;
  }
  return r;
}

// per-beam contribution to (dTr, H) -- OccGridMapUtil.h:76-98
struct Acc9 {
  float d0, d1, d2, h00, h11, h22, h01, h02, h12;
  __device__ __forceinline__ void zero() { d0 = d1 = d2 = h00 = h11 = h22 = h01 = h02 = h12 = 0.0f; }
};

// transform * currPoint with transform = Translation(ex,ey) * Rotation(theta): linear [c -s; s c]
template <int LAYOUT>
__device__ __forceinline__ BeamSample beam_fetch(const LevelView& L, float ex, float ey, float sinRot,
                                                 float cosRot, float px, float py, bool live = true) {
  const float tx = ex + (cosRot * px + (-sinRot) * py);
  const float ty = ey + (sinRot * px + cosRot * py);
  return sample_fetch<LAYOUT>(L, tx, ty, live);
}

__device__ __forceinline__ float beam_finish(const BeamSample& b, float sinRot, float cosRot, float px, float py,
                                             Acc9& a, BeamTerms* terms_out = nullptr) {
  const BeamTerms t = sample_finish(b);
  const float funVal = 1.0f - t.M;
  a.d0 += t.gx * funVal;
  a.d1 += t.gy * funVal;
  const float rotDeriv = ((-sinRot * px - cosRot * py) * t.gx + (cosRot * px - sinRot * py) * t.gy);  // :87
  a.d2 += rotDeriv * funVal;
  a.h00 += t.gx * t.gx;
  a.h11 += t.gy * t.gy;
  a.h22 += rotDeriv * rotDeriv;
  a.h01 += t.gx * t.gy;
  a.h02 += t.gx * rotDeriv;
  a.h12 += t.gy * rotDeriv;
  if (terms_out) *terms_out = t;
  return rotDeriv;
}

template <int LAYOUT>
__device__ __forceinline__ float beam_accumulate(const LevelView& L, float ex, float ey, float sinRot,
                                                 float cosRot, float px, float py, Acc9& a,
                                                 BeamTerms* terms_out = nullptr, bool live = true) {
  const BeamSample b = beam_fetch<LAYOUT>(L, ex, ey, sinRot, cosRot, px, py, live);
  return beam_finish(b, sinRot, cosRot, px, py, a, terms_out);
}

__device__ __forceinline__ float wave_allreduce(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ void wave_allreduce9(Acc9& a) {
  a.d0 = wave_allreduce(a.d0);
  a.d1 = wave_allreduce(a.d1);
  a.d2 = wave_allreduce(a.d2);
  a.h00 = wave_allreduce(a.h00);
  a.h11 = wave_allreduce(a.h11);
  a.h22 = wave_allreduce(a.h22);
  a.h01 = wave_allreduce(a.h01);
  a.h02 = wave_allreduce(a.h02);
  a.h12 = wave_allreduce(a.h12);
}

// team-wide totals: wave butterfly, then (WPS > 1) LDS staging of the per-wave
// partials; every thread of the team returns with identical bits.
template <int WPS>
__device__ __forceinline__ void team_allreduce9(Acc9& a, float (*red)[WPS][9], int buf, int wave_in_team,
                                                int lane) {
  wave_allreduce9(a);
  if (WPS > 1) {
    if (lane == 0) {
      float* r = red[buf][wave_in_team];
      r[0] = a.d0; r[1] = a.d1; r[2] = a.d2;
      r[3] = a.h00; r[4] = a.h11; r[5] = a.h22;
      r[6] = a.h01; r[7] = a.h02; r[8] = a.h12;
    }
    __syncthreads();
    float t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = red[buf][0][k];
#pragma unroll
    for (int w = 1; w < WPS; ++w) {
#pragma unroll
      for (int k = 0; k < 9; ++k) t[k] += red[buf][w][k];
    }
    a.d0 = t[0]; a.d1 = t[1]; a.d2 = t[2];
    a.h00 = t[3]; a.h11 = t[4]; a.h22 = t[5];
    a.h01 = t[6]; a.h02 = t[7]; a.h12 = t[8];
  }
}

// H.inverse() * dTr as Eigen evaluates it (LU/InverseImpl.h cofactors * invdet, then a
// coefficient-based product; 3-term sums are x0 + (x1 + x2)), ScanMatcher.h:201-217.
__device__ __forceinline__ void gn_solve_and_step(const Acc9& a, float& ex, float& ey, float& eth) {
  if ((a.h00 != 0.0f) && (a.h11 != 0.0f)) {
    // symmetric H: m(r,c)
    const float m00 = a.h00, m01 = a.h01, m02 = a.h02;
    const float m10 = a.h01, m11 = a.h11, m12 = a.h12;
    const float m20 = a.h02, m21 = a.h12, m22 = a.h22;
    // cofactor_3x3<i,j> = m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1), i1=(i+1)%3 ...
    const float c00 = m11 * m22 - m12 * m21;
    const float c10 = m21 * m02 - m22 * m01;
    const float c20 = m01 * m12 - m02 * m11;
    const float det = c00 * m00 + (c10 * m10 + c20 * m20);
    const float invdet = 1.0f / det;
    const float i00 = c00 * invdet, i01 = c10 * invdet, i02 = c20 * invdet;
    const float i10 = (m12 * m20 - m10 * m22) * invdet;  // cofactor<0,1>
    const float i11 = (m22 * m00 - m20 * m02) * invdet;  // cofactor<1,1>
    const float i12 = (m02 * m10 - m00 * m12) * invdet;  // cofactor<2,1>
    const float i20 = (m10 * m21 - m11 * m20) * invdet;  // cofactor<0,2>
    const float i21 = (m20 * m01 - m21 * m00) * invdet;  // cofactor<1,2>
    const float i22 = (m00 * m11 - m01 * m10) * invdet;  // cofactor<2,2>
    const float s0 = i00 * a.d0 + (i01 * a.d1 + i02 * a.d2);
    const float s1 = i10 * a.d0 + (i11 * a.d1 + i12 * a.d2);
    float s2 = i20 * a.d0 + (i21 * a.d1 + i22 * a.d2);
    if (s2 > 0.2f) {
      s2 = 0.2f;
    } else if (s2 < -0.2f) {
      s2 = -0.2f;
    }
    ex += s0;
    ey += s1;
    eth += s2;
  }
}

// One team (WPS wavefronts) per scan; SPB scans per workgroup (SPB > 1 only when WPS == 1,
// where no barrier is ever executed so the wavefronts of a block are fully independent).
//
// BPL > 0: "beams per lane" register-resident form.  A scan with n <= 64*WPS*BPL beams is loaded
// ONCE (coalesced float2, beam i -> slot i / (64*WPS) of lane i mod 64*WPS) and stays in VGPRs for
// all levels and GN steps; the beam loop is fully unrolled so the BPL texel gathers of a step are
// independent and issue back to back (latency hiding by ILP, not only by occupancy).  The
// per-lane summation order (ascending beam index) is the same as the memory loop's, so both
// forms produce identical bits.  Longer scans (or BPL == 0) take the memory loop.
template <int WPS, int SPB, int LAYOUT, int BPL>
__global__ void __launch_bounds__(64 * WPS * SPB, 4) gn_match_kernel(const MatchParams P) {
  static_assert(WPS == 1 || SPB == 1, "barrier-synchronised teams own their workgroup");
  constexpr int T = 64 * WPS;  // lanes per team
  __shared__ float red[2][WPS][9];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int team = wave / WPS;
  const int wit = wave - team * WPS;
  const int scan = blockIdx.x * SPB + team;
  if (scan >= P.batch) return;  // whole team exits together

  int beg = 0, n = P.shared_n;
  if (P.offsets) {
    beg = P.offsets[scan];
    n = P.offsets[scan + 1] - beg;
  }
  float pw0 = P.begin_world[3 * scan + 0];
  float pw1 = P.begin_world[3 * scan + 1];
  float pw2 = P.begin_world[3 * scan + 2];
  if (n == 0) {  // ScanMatcher.h:68,189: pose passes through, cov untouched
    if (lane == 0 && wit == 0) {
      P.out_pose[3 * scan + 0] = pw0;
      P.out_pose[3 * scan + 1] = pw1;
      P.out_pose[3 * scan + 2] = pw2;
    }
    return;
  }
  const float2* __restrict__ pts = P.pts + beg;
  const int tid_in_team = wit * 64 + lane;
  constexpr int NREG = BPL > 0 ? BPL : 1;
  float2 pt[NREG];
  unsigned live_mask = 0;
  const bool in_regs = BPL > 0 && n <= T * BPL;  // team-uniform
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int i = tid_in_team + k * T;
      const bool live = i < n;
      pt[k] = live ? pts[i] : make_float2(0.0f, 0.0f);
      live_mask |= (live ? 1u : 0u) << k;
    }
  }
  Acc9 acc;
  acc.zero();
  int buf = 0;
  int step = 0;
  float reg_scale = 1.0f;  // scale the register-resident endpoints currently carry
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);  // getMapCoordsPose, GridMapBase.h:235-239
    eth = pw2;
    const float ps = L.pt_scale;
    if (in_regs) {
      // DataContainer::setFrom(scan, 2^-level): rescale IN PLACE when the level changes.  All
      // factors are powers of two, so p*2^-a*2^(a-b) == p*2^-b bit for bit, and no second
      // register copy of the scan is kept alive across the GN steps.
      const float ratio = ps / reg_scale;
      reg_scale = ps;
#pragma unroll
      for (int k = 0; k < NREG; ++k) {
        pt[k].x *= ratio;
        pt[k].y *= ratio;
      }
    }
    for (int it = 0; it < L.gn_steps; ++it) {
      float sinRot, cosRot;
      sincos_f32(eth, sinRot, cosRot);
      acc.zero();
      if (in_regs) {
        // chunks of kUnroll beams: issue all gathers of a chunk, then consume them in beam order
        // (the accumulation order stays k = 0, 1, 2 ... so the bits equal the memory loop's)
#pragma unroll
        for (int k0 = 0; k0 < NREG; k0 += kUnroll) {
          BeamSample smp[kUnroll];
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            if (k0 + u < NREG)
              smp[u] = beam_fetch<LAYOUT>(L, ex, ey, sinRot, cosRot, pt[k0 + u].x, pt[k0 + u].y,
                                          (live_mask >> (k0 + u)) & 1u);
          }
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            if (k0 + u < NREG) beam_finish(smp[u], sinRot, cosRot, pt[k0 + u].x, pt[k0 + u].y, acc);
          }
          // Pin the chunk: the accumulators must be final here ("+v") and no later gather may be
          // hoisted above this point ("memory").  Without it the compiler issues all BPL gathers
          // first and spills their results; with it at most kUnroll texels are in flight per lane.
          asm volatile(""
                       : "+v"(acc.d0), "+v"(acc.d1), "+v"(acc.d2), "+v"(acc.h00), "+v"(acc.h11), "+v"(acc.h22),
                         "+v"(acc.h01), "+v"(acc.h02), "+v"(acc.h12)
                       :
                       : "memory");
        }
      } else {
        for (int i = tid_in_team; i < n; i += T) {
          const float2 p = pts[i];
          beam_accumulate<LAYOUT>(L, ex, ey, sinRot, cosRot, p.x * ps, p.y * ps, acc);
        }
      }
      team_allreduce9<WPS>(acc, red, buf, wit, lane);
      buf ^= 1;
      gn_solve_and_step(acc, ex, ey, eth);
      if (P.trace) {  // kernel-uniform; only the single-scan hook path sets it
        if (scan == 0 && lane == 0 && wit == 0) {
          float* t = P.trace + 12 * step;
          t[0] = ex; t[1] = ey; t[2] = eth;
          t[3] = acc.h00; t[4] = acc.h01; t[5] = acc.h02;
          t[6] = acc.h01; t[7] = acc.h11; t[8] = acc.h12;
          t[9] = acc.h02; t[10] = acc.h12; t[11] = acc.h22;
        }
        ++step;
      }
    }
    eth = normalize_angle(eth);                     // ScanMatcher.h:170
    affine_apply(L.worldTmap, ex, ey, pw0, pw1);    // getWorldCoordsPose, :186
    pw2 = eth;
  }
  if (lane == 0 && wit == 0) {
    P.out_pose[3 * scan + 0] = pw0;
    P.out_pose[3 * scan + 1] = pw1;
    P.out_pose[3 * scan + 2] = pw2;
    if (P.out_cov) {  // covMatrix = H of the last evaluation (ScanMatcher.h:184), column major
      float* c = P.out_cov + 9 * scan;
      c[0] = acc.h00; c[1] = acc.h01; c[2] = acc.h02;
      c[3] = acc.h01; c[4] = acc.h11; c[5] = acc.h12;
      c[6] = acc.h02; c[7] = acc.h12; c[8] = acc.h22;
    }
  }
}

// ---- parity / debug kernels: one evaluation at a given map-frame pose ------------
// H, dTr of one getCompleteHessianDerivs call (same device functions as the matcher)
template <int LAYOUT>
__global__ void __launch_bounds__(1024) gn_eval_kernel(const LevelView L, const float2* __restrict__ pts,
                                                      int n, float ex, float ey, float eth,
                                                      float* out12 /* H[9] col-major, dTr[3] */) {
  __shared__ float red[2][16][9];
  const int lane = threadIdx.x & 63;
  const int wit = threadIdx.x >> 6;
  float sinRot, cosRot;
  sincos_f32(eth, sinRot, cosRot);
  Acc9 acc;
  acc.zero();
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float2 p = pts[i];
    beam_accumulate<LAYOUT>(L, ex, ey, sinRot, cosRot, p.x, p.y, acc);
  }
  team_allreduce9<16>(acc, red, 0, wit, lane);
  if (threadIdx.x == 0) {
    out12[0] = acc.h00; out12[1] = acc.h01; out12[2] = acc.h02;
    out12[3] = acc.h01; out12[4] = acc.h11; out12[5] = acc.h12;
    out12[6] = acc.h02; out12[7] = acc.h12; out12[8] = acc.h22;
    out12[9] = acc.d0; out12[10] = acc.d1; out12[11] = acc.d2;
  }
}

// per-beam M, dM/dx, dM/dy, rotDeriv
template <int LAYOUT>
__global__ void gn_beam_terms_kernel(const LevelView L, const float2* __restrict__ pts, int n, float ex,
                                     float ey, float eth, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sinRot, cosRot;
  sincos_f32(eth, sinRot, cosRot);
  Acc9 acc;
  acc.zero();
  BeamTerms t;
  const float2 p = pts[i];
  const float rd = beam_accumulate<LAYOUT>(L, ex, ey, sinRot, cosRot, p.x, p.y, acc, &t);
  out[i] = make_float4(t.M, t.gx, t.gy, rd);
}

// device sin/cos sweep for the parity tests
__global__ void sincos_debug_kernel(const float* __restrict__ x, int n, float* __restrict__ s,
                                    float* __restrict__ c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sincos_f32(x[i], s[i], c[i]);
}

}  // namespace hsm
