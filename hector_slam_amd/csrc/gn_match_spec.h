// gn_match_spec.h -- ONE scan per workgroup in the reference's summation order, with the nine chains OFF the critical path
// (round 6): segments of every chain run in parallel from speculated carries and are stitched by the exact shift rule of
// spec_chain.h; bit-identical to the literal chains by construction (a segment the rule does not accept is re-run literally).
//
// What it replaces: gn_match_exact_dense_kernel (round 5) ran a 16 384-beam scan of configs[4] at the floor of the literal
// chain -- 16 384 x 8.5 cycles per Gauss-Newton step, 0.92 ms per match -- with fifteen of its sixteen wavefronts waiting for
// one.  Here a GN step is
//   A  production   all 1024 lanes: an item = four consecutive beams (endpoints -> rotate -> bilinear texel -> the nine products of
//                   OccGridMapUtil.h:83-97, each rounded like the reference's), stored as one float4 per chain into a scratch
//                   block (HBM / L2) laid out [chain][chunk][lane] -- consecutive items belong to consecutive LANES of the wavefront
//                   that will add them, so both the stores here and the loads of phase C are coalesced -- and summed (fp32, LDS
//                   atomics, any order) into one number per chain and lane: the material for the candidate carries;
//   C  run          wavefront c = chain c, lane L owns G consecutive segments of m beams (spec::plan): an exclusive fp32 prefix of the
//                   lane sums is the candidate carry of its first segment; the lane runs the LITERAL fp32 loop through its segments
//                   (a segment's candidate is the running value of the one before) and keeps, per segment, candidate, end value and
//                   the admissible shifts (five independent operations per addition: spec_chain.h) in registers;
//   D  stitch       the same wavefront: hypothesise that every segment from the frontier on accepts its shift (an exclusive scan
//                   over the lanes, fp64, every addition checked for exactness), find the FIRST one that does not (one ballot),
//                   re-run that segment literally from its true carry (m dependent additions fed by v_readlane), move the frontier
//                   behind it, repeat -- real chains need 10-16 re-runs per chain whatever their length (they are the places where
//                   the running sum crosses a power of two; tools/study/spec_chain_stats.py);
//   then the nine totals go through LDS to every lane, which solves the 3x3 system redundantly (as every form does).
// The arithmetic that decides the RESULT is only ever the literal fp32 addition, in beam order, and the shift rule's exact
// additions; candidates, summaries and the order of the atomics only decide how many segments are re-run.
// Host model of phases C and D (same header, same arithmetic): tests/cpp/spec_chain_model.cpp, speculative_wave().
#pragma once
#include "gn_match.h"
#include "spec_chain.h"

namespace hsm {

// float4s of scratch a scan of at most n_bound beams needs, whatever its own length (per_lane(n) <= span(n) + 32)
__host__ __device__ inline size_t spec_scratch_float4s_bound(int n_bound) {
  int span = (n_bound + 63) / 64;
  span = ((span + 3) & ~3) + 32;
  if (span < 40) span = 40;
  return (size_t)9 * 64 * (size_t)(span / 4);
}

__device__ __forceinline__ float spec_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <int LAYOUT>
__global__ void __launch_bounds__(1024) gn_match_spec_kernel(const MatchParams P) {
  __shared__ float lane_sum[9 * 64];
  __shared__ float totals[9];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int scan = (int)blockIdx.x;
  int beg = 0, n = P.shared_n;
  if (P.offsets) {
    beg = P.offsets[scan];
    n = P.offsets[scan + 1] - beg;
  }
  float pw0, pw1, pw2;
  if (P.begin_world) {
    pw0 = P.begin_world[3 * scan + 0];
    pw1 = P.begin_world[3 * scan + 1];
    pw2 = P.begin_world[3 * scan + 2];
  } else {
    pw0 = P.begin_inline[0];
    pw1 = P.begin_inline[1];
    pw2 = P.begin_inline[2];
  }
  if (n == 0) {  // ScanMatcher.h:68,189
    if (tid == 0) {
      P.out_pose[3 * scan + 0] = pw0;
      P.out_pose[3 * scan + 1] = pw1;
      P.out_pose[3 * scan + 2] = pw2;
      if (scan == 0) publish_done(P);
    }
    return;
  }
  const spec::Plan pl = spec::plan(n);
  const int G = pl.G, m = pl.m;
  const int per_lane = G * m;     // beams a lane owns
  const int CPL = per_lane >> 2;  // float4 chunks per lane and chain
  const int NS = pl.lanes * G;    // segments (the last lane's may be partly or wholly padding)
  if ((size_t)9 * 64 * (size_t)CPL > (size_t)P.spec_stride) {  // cannot happen when the host sized the scratch from a true bound
    if (tid == 0) {                                           // (n_bound): refuse loudly instead of writing outside the block
      const float nan = __uint_as_float(0x7fc00000u);
      P.out_pose[3 * scan + 0] = nan;
      P.out_pose[3 * scan + 1] = nan;
      P.out_pose[3 * scan + 2] = nan;
      if (scan == 0) publish_done(P);
    }
    return;
  }
  const float2* __restrict__ pts = P.pts + beg;
  f4v* scratch = reinterpret_cast<f4v*>(P.spec_scratch) + (size_t)scan * P.spec_stride;
  const int items = CPL * 64;  // (q, L), L fastest; items of lanes >= pl.lanes hold no beam and are skipped
  const int rounds = (items + 1023) >> 10;
  Acc9 acc;
  acc.zero();
  int step = 0;
  unsigned st_bound = 0, st_shift = 0, st_rerun = 0;
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);
    eth = pw2;
    const float ps = L.pt_scale;
    const int gn_steps = L.gn_steps;
    const LevelRegs R = level_regs<LAYOUT>(L);
    for (int it = 0; it < gn_steps; ++it) {
      float sinRot, cosRot;
      sincos_f32(eth, sinRot, cosRot);
      const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
      unsigned long long ts[5];
      const bool probe = P.clock_probe != nullptr && scan == 0 && tid == 0;  // (wave 0 takes part in every phase)
      if (probe) ts[0] = __builtin_readcyclecounter();
      if (tid < 9 * 64) lane_sum[tid] = 0.0f;
      __syncthreads();
      // ---- A: production ------------------------------------------------------------------------------------------------
      for (int r = 0; r < rounds; ++r) {
        const int w = (r << 10) + tid;
        const int q = w >> 6, Lw = w & 63;
        if (w < items && Lw < pl.lanes) {
          const int i0 = Lw * per_lane + (q << 2);  // first of the item's four beams
          float pr[4][9];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int i = i0 + b;
            // beams beyond n inside the last lane's span: an endpoint outside any map -> exact +-0 products (every form's padding)
            const float2 pq = i < n ? pts[i] : make_float2(1.0e30f, 1.0e30f);
            BeamRot rot;
            const BeamSample s = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{pq.x * ps, pq.y * ps}, rot);
            beam_products(s, rot, pr[b]);
          }
          f4v* dst = scratch + (size_t)q * 64 + Lw;
#pragma unroll
          for (int c = 0; c < 9; ++c) {
            dst[(size_t)c * CPL * 64] = f4v{pr[0][c], pr[1][c], pr[2][c], pr[3][c]};
            atomicAdd(&lane_sum[c * 64 + Lw], (pr[0][c] + pr[1][c]) + (pr[2][c] + pr[3][c]));
          }
        }
      }
      __syncthreads();
      if (probe) ts[1] = __builtin_readcyclecounter();
      if (wave < 9) {
        const int c = wave;
        const bool live = lane < pl.lanes;
        // ---- candidates: exclusive fp32 prefix of the lane sums ------------------------------------------------------------
        float incl = live ? lane_sum[c * 64 + lane] : 0.0f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float up = __shfl_up(incl, o);
          if (lane >= o) incl += up;
        }
        float first = __shfl_up(incl, 1);
        if (lane == 0) first = 0.0f;  // the chain's own start: +0, and that IS the true carry
        // ---- C: the lane's continuous literal run with its shift summaries --------------------------------------------------------
        float s_cand[8], s_fin[8], s_lo[8], s_hi[8], s_inv[8];
        {
          float run = first;
          const f4v* src = scratch + (size_t)c * CPL * 64 + lane;
          const int cps = m >> 2;  // chunks per segment
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            s_cand[g] = run;
            spec::SegSummary S;
            S.reset(run);
            if (g < G && live) {
              f4v nx = src[(size_t)(g * cps) * 64];
              for (int qq = 0; qq < cps; ++qq) {
                const f4v v = nx;
                if (qq + 1 < cps) nx = src[(size_t)(g * cps + qq + 1) * 64];
                float s;
                s = run; run = s + v.x; spec::seg_step(S, s, v.x, run);
                s = run; run = s + v.y; spec::seg_step(S, s, v.y, run);
                s = run; run = s + v.z; spec::seg_step(S, s, v.z, run);
                s = run; run = s + v.w; spec::seg_step(S, s, v.w, run);
              }
            }
            const spec::SegShifts sh = spec::seg_shifts(S);
            s_fin[g] = run;
            s_lo[g] = sh.lo;
            s_hi[g] = sh.hi;
            s_inv[g] = sh.unit != 0.0f ? 1.0f / sh.unit : 0.0f;  // a power of two: exact
          }
        }
        if (probe) ts[2] = __builtin_readcyclecounter();
        // ---- D: stitch (frontier loop) --------------------------------------------------------------------------------------
        // e = (end value of the lane before) - (this lane's first candidate): what the shift changes by across the lane boundary
        float e_in;
        bool e_exact;
        {
          float last_fin = s_fin[0];
#pragma unroll
          for (int g = 1; g < 8; ++g) last_fin = (g == G - 1) ? s_fin[g] : last_fin;
          const float pf = __shfl_up(last_fin, 1);
          e_in = spec::seg_delta(pf, s_cand[0], &e_exact);
        }
        int F = 0;       // frontier: the first segment whose carry is not settled (wave-uniform)
        float t = 0.0f;  // the true carry into segment F (wave-uniform)
        float total = 0.0f;
        for (;;) {
          if (F >= NS) {
            total = t;
            break;
          }
          const int LF = F / G, gF = F - LF * G;
          // shift of every lane's pending segments under the hypothesis: lane LF: t - cand[gF]; lane L > LF: + e of every boundary between
          float candF = s_cand[0];
#pragma unroll
          for (int g = 1; g < 8; ++g) candF = (g == gF) ? s_cand[g] : candF;
          bool d0_exact;
          const float d0 = spec::seg_delta(t, spec_readlane(candF, LF), &d0_exact);
          double v = lane == LF ? (double)d0 : (lane > LF ? (double)e_in : 0.0);
          bool okv = lane == LF ? d0_exact : (lane > LF ? e_exact : true);
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {  // inclusive scan; a TwoSum error or an inexact term poisons everything behind it
            const double uv = __shfl_up(v, o);
            const int uo = __shfl_up((int)okv, o);
            if (lane >= o) {
              const double sum = v + uv;
              const double bb = sum - v;
              const double err = (v - (sum - bb)) + (uv - bb);
              okv = okv && (uo != 0) && err == 0.0;
              v = sum;
            }
          }
          const float d = (float)v;
          const bool d_ok = okv && (double)d == v;
          // first pending segment of this lane that does not accept d
          int g_fail = 8;
#pragma unroll
          for (int g = 7; g >= 0; --g) {
            const bool pending = g < G && (lane > LF || (lane == LF && g >= gF)) && lane * G + g < NS;
            bool ok = d_ok && (d == 0.0f);
            if (d_ok && !ok && s_inv[g] != 0.0f && d >= s_lo[g] && d <= s_hi[g]) {
              const float qa = spec::u2f(spec::f2u(d * s_inv[g]) & 0x7fffffffu);
              ok = qa >= 8388608.0f ? qa < 3.0e38f : ((qa + 8388608.0f) - 8388608.0f) == qa;
            }
            if (pending && !ok) g_fail = g;
          }
          const unsigned long long failing = __builtin_amdgcn_ballot_w64(g_fail < 8);
          if (failing == 0ull) {  // everything behind the frontier accepted: the chain's end is the last segment's end, shifted
            const int Ll = (NS - 1) / G, gl = (NS - 1) - Ll * G;
            float fl = s_fin[0];
#pragma unroll
            for (int g = 1; g < 8; ++g) fl = (g == gl) ? s_fin[g] : fl;
            total = spec_readlane(fl + d, Ll);
            if (P.spec_stats) st_bound += (unsigned)(NS - F), st_shift += (unsigned)(NS - F);
            break;
          }
          const int Lx = (int)__builtin_ctzll(failing);
          const int gx = __builtin_amdgcn_readlane(g_fail, Lx);
          const int X = Lx * G + gx;  // the first segment that must be re-run
          // its true carry: t if it is the frontier itself, else the end of the segment before it, shifted
          float t_start = t;
          if (X != F) {
            const int gp = gx > 0 ? gx - 1 : G - 1;
            float pf = s_fin[0];
#pragma unroll
            for (int g = 1; g < 8; ++g) pf = (g == gp) ? s_fin[g] : pf;
            t_start = spec_readlane(pf + d, gx > 0 ? Lx : Lx - 1);
          }
          if (P.spec_stats) st_bound += (unsigned)(X - F + 1), st_shift += (unsigned)(X - F), ++st_rerun;
          // re-run segment X literally: lane qq fetches chunk qq of the segment, then m dependent additions fed by v_readlane
          {
            const int cps = m >> 2;
            const f4v* src = scratch + ((size_t)c * CPL + (size_t)gx * cps) * 64 + Lx;
            float run = t_start;
            for (int q0 = 0; q0 < cps; q0 += 64) {
              const int qn = min(64, cps - q0);
              f4v vv = f4v{0.0f, 0.0f, 0.0f, 0.0f};
              if (lane < qn) vv = src[(size_t)(q0 + lane) * 64];
              for (int qq = 0; qq < qn; ++qq) {
                const int ql = __builtin_amdgcn_readfirstlane(qq);
                run += spec_readlane(vv.x, ql);
                run += spec_readlane(vv.y, ql);
                run += spec_readlane(vv.z, ql);
                run += spec_readlane(vv.w, ql);
              }
            }
            t = run;
          }
          F = X + 1;
        }
        if (lane == 0) totals[c] = total;
        if (probe) ts[3] = __builtin_readcyclecounter();
      }
      __syncthreads();
      if (probe) {
        ts[4] = __builtin_readcyclecounter();
        for (int k = 0; k < 5; ++k) P.clock_probe[k] = ts[k];
      }
      acc.d01 = f2{totals[0], totals[1]}; acc.d2 = totals[2];
      acc.hd = f2{totals[3], totals[4]}; acc.h22 = totals[5];
      acc.h01 = totals[6]; acc.hr = f2{totals[7], totals[8]};
      gn_solve_and_step(acc, ex, ey, eth);
      if (P.trace) {  // kernel-uniform; only the single-scan hook path sets it
        if (scan == 0 && tid == 0) {
          float* tr = P.trace + 12 * step;
          tr[0] = ex; tr[1] = ey; tr[2] = eth;
          tr[3] = acc.hd.x; tr[4] = acc.h01; tr[5] = acc.hr.x;
          tr[6] = acc.h01; tr[7] = acc.hd.y; tr[8] = acc.hr.y;
          tr[9] = acc.hr.x; tr[10] = acc.hr.y; tr[11] = acc.h22;
        }
        ++step;
      }
      // (the next step zeroes lane_sum and rewrites the scratch block behind its first barrier; totals[] is rewritten behind its
      // second: every read of this step lies in front of them)
    }
    eth = normalize_angle(eth);
    affine_apply(L.worldTmap, ex, ey, pw0, pw1);
    pw2 = eth;
  }
  if (P.spec_stats && wave < 9 && lane == 0) {
    atomicAdd(&P.spec_stats->boundaries, (unsigned long long)st_bound);
    atomicAdd(&P.spec_stats->shifted, (unsigned long long)st_shift);
    atomicAdd(&P.spec_stats->rerun, (unsigned long long)st_rerun);
  }
  if (tid == 0) {
    P.out_pose[3 * scan + 0] = pw0;
    P.out_pose[3 * scan + 1] = pw1;
    P.out_pose[3 * scan + 2] = pw2;
    if (P.out_cov) {  // covMatrix = H of the last evaluation (ScanMatcher.h:184), column major
      float* c = P.out_cov + 9 * scan;
      c[0] = acc.hd.x; c[1] = acc.h01; c[2] = acc.hr.x;
      c[3] = acc.h01; c[4] = acc.hd.y; c[5] = acc.hr.y;
      c[6] = acc.hr.x; c[7] = acc.hr.y; c[8] = acc.h22;
    }
    if (scan == 0) publish_done(P);
  }
}

// ---- the same idea for ONE scan of the node's size (<= 2048 beams), everything on chip (round 6) ------------------------------------
// The ROS node's matchData is a single 1081-beam scan: 14 GN steps x 1081 dependent additions x 8.5 cycles = 54 of the 87 us the
// exact team form takes.  Here the products of a step never leave the CU: 16 wavefronts produce them into LDS (endpoints resident in
// registers across all levels and steps, read ONCE -- straight from pinned host memory -- like the team form's), then wavefront c
// = chain c: lane L loads ITS segment (m <= 32 consecutive products, spec::plan with G = 1) into registers, sums it for the
// candidate carries (fp32 prefix over the lanes), runs the literal loop from its candidate with the shift summary, and the frontier
// loop stitches: the boundary offsets e_L = f_{L-1} - c_L are prefix-summed ONCE per step (fp64, exact or flagged), so a pass of the
// loop is a handful of per-lane operations, one ballot, and -- for the first segment that does not accept -- m dependent additions
// fed by v_readlane from the owner lane's registers.  No global memory behind the texel gather.
constexpr int kSpec1MaxBeams = 2048;
constexpr int kSpec1Row = 64 * 33;  // floats per chain in LDS: 64 lanes x (m + 1), m <= 32 (odd stride: conflict-free column reads)

template <int LAYOUT>
__global__ void __launch_bounds__(1024) gn_match_spec1_kernel(const MatchParams P) {
  __shared__ float prod[9 * kSpec1Row];
  __shared__ float totals[9];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int scan = (int)blockIdx.x;
  int beg = 0, n = P.shared_n;
  if (P.offsets) {
    beg = P.offsets[scan];
    n = P.offsets[scan + 1] - beg;
  }
  float pw0, pw1, pw2;
  if (P.begin_world) {
    pw0 = P.begin_world[3 * scan + 0];
    pw1 = P.begin_world[3 * scan + 1];
    pw2 = P.begin_world[3 * scan + 2];
  } else {
    pw0 = P.begin_inline[0];
    pw1 = P.begin_inline[1];
    pw2 = P.begin_inline[2];
  }
  if (n == 0 || n > kSpec1MaxBeams) {  // empty: ScanMatcher.h:68,189; too long: the host never sends it here -- refuse loudly
    if (tid == 0) {
      const float nan = __uint_as_float(0x7fc00000u);
      P.out_pose[3 * scan + 0] = n == 0 ? pw0 : nan;
      P.out_pose[3 * scan + 1] = n == 0 ? pw1 : nan;
      P.out_pose[3 * scan + 2] = n == 0 ? pw2 : nan;
      if (scan == 0) publish_done(P);
    }
    return;
  }
  const spec::Plan pl = spec::plan(n);  // G == 1 for n <= 2048
  const int m = pl.m, lanes = pl.lanes, ms = m + 1;
  const float2* __restrict__ pts = P.pts + beg;
  // this lane's (up to two) beams: endpoint, and where its products go -- fixed for the whole match
  float2 q0 = make_float2(1.0e30f, 1.0e30f), q1 = q0;
  int off0 = -1, off1 = -1;
  {
    const int i0 = tid, i1 = tid + 1024;
    if (i0 < n) {
      q0 = pts[i0];
      const int L = i0 / m;
      off0 = L * ms + (i0 - L * m);
    }
    if (i1 < n) {
      q1 = pts[i1];
      const int L = i1 / m;
      off1 = L * ms + (i1 - L * m);
    }
  }
  // the padding of the last segment: +0 products, written once (no beam ever lands there)
  for (int i = n + tid; i < lanes * m; i += 1024) {
    const int L = i / m, o = L * ms + (i - L * m);
#pragma unroll
    for (int c = 0; c < 9; ++c) prod[c * kSpec1Row + o] = 0.0f;
  }
  Acc9 acc;
  acc.zero();
  int step = 0;
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);
    eth = pw2;
    const float ps = L.pt_scale;
    const int gn_steps = L.gn_steps;
    const LevelRegs R = level_regs<LAYOUT>(L);
    for (int it = 0; it < gn_steps; ++it) {
      float sinRot, cosRot;
      sincos_f32(eth, sinRot, cosRot);
      const f2 e2 = step_origin(ex, ey), cs = f2{cosRot, sinRot}, sc = f2{sinRot, cosRot};
      const bool probe = P.clock_probe != nullptr && tid == 0;
      unsigned long long ts[8];
      int iters = 0;
      if (probe) ts[0] = __builtin_readcyclecounter();
      // ---- production: the nine products of this lane's beams into LDS ---------------------------------------------------
      if (off0 >= 0) {
        BeamRot rot;
        const BeamSample b = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{q0.x * ps, q0.y * ps}, rot);
        float pr[9];
        beam_products(b, rot, pr);
#pragma unroll
        for (int c = 0; c < 9; ++c) prod[c * kSpec1Row + off0] = pr[c];
      }
      if (off1 >= 0) {
        BeamRot rot;
        const BeamSample b = beam_fetch<LAYOUT>(R, e2, cs, sc, f2{q1.x * ps, q1.y * ps}, rot);
        float pr[9];
        beam_products(b, rot, pr);
#pragma unroll
        for (int c = 0; c < 9; ++c) prod[c * kSpec1Row + off1] = pr[c];
      }
      __syncthreads();
      if (probe) ts[1] = __builtin_readcyclecounter();
      if (wave < 9) {
        const int c = wave;
        const bool live = lane < lanes;
        // ---- this lane's segment into registers; its fp32 sum for the candidate carries --------------------------------------
        float x[32];
        const float* src = prod + c * kSpec1Row + lane * ms;
#pragma unroll
        for (int k = 0; k < 32; ++k) x[k] = (live && k < m) ? src[k] : 0.0f;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; k += 4) s0 += x[k], s1 += x[k + 1], s2 += x[k + 2], s3 += x[k + 3];
        float incl = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float up = __shfl_up(incl, o);
          if (lane >= o) incl += up;
        }
        float cand = __shfl_up(incl, 1);
        if (lane == 0) cand = 0.0f;  // the chain's own start: +0, the true carry
        if (probe) ts[2] = __builtin_readcyclecounter();
        // ---- the literal run from the candidate with its shift summary ---------------------------------------------------------
        float run = cand;
        spec::SegSummary S;
        S.reset(run);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          if (k < m) {  // (wave-uniform)
            const float before = run;
            run = before + x[k];
            spec::seg_step(S, before, x[k], run);
          }
        }
        const float fin = run;
        const spec::SegShifts sh = spec::seg_shifts(S);
        const float inv = sh.unit != 0.0f ? 1.0f / sh.unit : 0.0f;  // a power of two: exact
        // ---- boundary offsets e_L = f_{L-1} - c_L, prefix-summed once: E_L (fp64), and how many boundaries / additions up to L
        // were not exact (a segment behind one of those, counted from the frontier, is re-run)
        if (probe) ts[3] = __builtin_readcyclecounter();
        bool e_exact = true;
        float e_in = 0.0f;
        {
          const float pf = __shfl_up(fin, 1);
          if (lane > 0) e_in = spec::seg_delta(pf, cand, &e_exact);
        }
        double E = (double)e_in;
        int nbad = e_exact ? 0 : 1;  // -> boundaries up to this lane whose offset is not exact (counted from the frontier below)
        int pois = 0;                // -> an addition of the scan itself was not exact: this lane's E is not to be used at all
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const double uE = __shfl_up(E, o);
          const int ub = __shfl_up(nbad, o);
          const int up = __shfl_up(pois, o);
          if (lane >= o) {
            const double sum = E + uE;
            const double bb = sum - E;
            const double err = (E - (sum - bb)) + (uE - bb);
            nbad += ub;
            pois |= up | (err != 0.0 ? 1 : 0);
            E = sum;
          }
        }
        // ---- the frontier loop ---------------------------------------------------------------------------------------------------
        if (probe) ts[4] = __builtin_readcyclecounter();
        int F = 0;
        float t = 0.0f;
        float total = 0.0f;
        for (;;) {
          ++iters;
          if (F >= lanes) {
            total = t;
            break;
          }
          bool d0_exact;
          const float d0 = spec::seg_delta(t, spec_readlane(cand, F), &d0_exact);
          const double EF = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(E), F), __builtin_amdgcn_readlane(__double2loint(E), F));
          const int bF = __builtin_amdgcn_readlane(nbad, F);
          const int pF = __builtin_amdgcn_readlane(pois, F);
          // shift of lane L >= F under the hypothesis that everything from F on accepts: d0 + (E_L - E_F), every operation checked
          double v;
          bool okv = d0_exact && nbad == bF && pois == 0 && pF == 0;
          {
            const double a = E, b = -EF;
            const double r = a + b;
            const double bb = r - a;
            const double err = (a - (r - bb)) + (b - bb);
            okv = okv && err == 0.0;
            const double a2 = r, b2 = (double)d0;
            v = a2 + b2;
            const double bb2 = v - a2;
            const double err2 = (a2 - (v - bb2)) + (b2 - bb2);
            okv = okv && err2 == 0.0;
          }
          const float d = (float)v;
          const bool d_ok = okv && (double)d == v;
          bool ok = d_ok && d == 0.0f;
          if (d_ok && !ok && inv != 0.0f && d >= sh.lo && d <= sh.hi) {
            const float qa = spec::u2f(spec::f2u(d * inv) & 0x7fffffffu);
            ok = qa >= 8388608.0f ? qa < 3.0e38f : ((qa + 8388608.0f) - 8388608.0f) == qa;
          }
          const bool pending = lane >= F && lane < lanes;
          const unsigned long long failing = __builtin_amdgcn_ballot_w64(pending && !ok);
          const float shifted_end = fin + d;  // this segment's true end value if it (and everything before it) accepted
          if (failing == 0ull) {
            total = spec_readlane(shifted_end, lanes - 1);
            break;
          }
          const int X = (int)__builtin_ctzll(failing);
          float rr = X == F ? t : spec_readlane(shifted_end, X - 1);
          // re-run segment X literally: its products sit in lane X's registers
#pragma unroll
          for (int k = 0; k < 32; ++k)
            if (k < m) rr += spec_readlane(x[k], X);
          t = rr;
          F = X + 1;
        }
        if (lane == 0) totals[c] = total;
        if (probe) ts[5] = __builtin_readcyclecounter();
      }
      __syncthreads();
      if (probe) {
        ts[6] = __builtin_readcyclecounter();
        for (int kk = 0; kk < 7; ++kk) P.clock_probe[kk] = ts[kk];
        P.clock_probe[7] = (unsigned long long)iters;
      }
      acc.d01 = f2{totals[0], totals[1]}; acc.d2 = totals[2];
      acc.hd = f2{totals[3], totals[4]}; acc.h22 = totals[5];
      acc.h01 = totals[6]; acc.hr = f2{totals[7], totals[8]};
      gn_solve_and_step(acc, ex, ey, eth);
      if (P.trace) {  // kernel-uniform; only the single-scan hook path sets it
        if (scan == 0 && tid == 0) {
          float* tr = P.trace + 12 * step;
          tr[0] = ex; tr[1] = ey; tr[2] = eth;
          tr[3] = acc.hd.x; tr[4] = acc.h01; tr[5] = acc.hr.x;
          tr[6] = acc.h01; tr[7] = acc.hd.y; tr[8] = acc.hr.y;
          tr[9] = acc.hr.x; tr[10] = acc.hr.y; tr[11] = acc.h22;
        }
        ++step;
      }
      // (the next step's production rewrites prod[] behind the barrier above: the chain wavefronts have had their segments in
      // registers since before the frontier loop; totals[] is rewritten behind the next production barrier)
    }
    eth = normalize_angle(eth);
    affine_apply(L.worldTmap, ex, ey, pw0, pw1);
    pw2 = eth;
  }
  if (tid == 0) {
    P.out_pose[3 * scan + 0] = pw0;
    P.out_pose[3 * scan + 1] = pw1;
    P.out_pose[3 * scan + 2] = pw2;
    if (P.out_cov) {  // covMatrix = H of the last evaluation (ScanMatcher.h:184), column major
      float* cc = P.out_cov + 9 * scan;
      cc[0] = acc.hd.x; cc[1] = acc.h01; cc[2] = acc.hr.x;
      cc[3] = acc.h01; cc[4] = acc.hd.y; cc[5] = acc.hr.y;
      cc[6] = acc.hr.x; cc[7] = acc.hr.y; cc[8] = acc.h22;
    }
    if (scan == 0) publish_done(P);
  }
}

}  // namespace hsm
