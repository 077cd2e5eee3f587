// gn_match_exact.h -- HSM_PARITY_EXACT for batches: the exact-order matcher with the texel cache (round 3).
//
// What it computes: the same coarse-to-fine Gauss-Newton match as gn_match.h, with the nine sums of
// OccGridMapUtil::getCompleteHessianDerivs (HSL/map/OccGridMapUtil.h:76-98) accumulated in the REFERENCE's order --
// one fp32 chain per term, beam 0 .. n-1 -- so H, dTr, every GN step (ScanMatcher.h:194-221), the pose and the
// covariance are bit-identical to the reference CPU matcher.
//
// How (measurements, what-if builds and the variants that lost: profiles/r03/README.md):
//   * one wavefront per scan, NS scans per workgroup (4 as launched: one wavefront per SIMD and workgroup, four independent
//     workgroups per CU); every wavefront is a PRODUCER with gn_match_cached_kernel's machinery: the last texel + byte
//     offset of its first BPC beams per lane stay in VGPRs (the other rows gather in every step), exec-masked inline-asm
//     gathers issued one beam ahead with counted s_waitcnt, endpoints in LDS (the first rows in VGPRs where the LDS share
//     of 16 scans per CU does not hold all rows next to the stage), the first GN step peeled (endpoints straight from
//     their load registers while they stream in);
//   * per ROUND k (beam k of every lane = beams 64k .. 64k+63 of the scan) a producer stages the NINE products of
//     :83-97 (each one rounding, like the reference's) in LDS rows, written with ds_write_addtid_b32;
//   * the chains: 9 NS sequential sums per workgroup, 64 additions each per round.  A chain JOB is one wavefront whose
//     lane l runs chain l of the round: 64 dependent v_add_f32 fed by ds_read_b128 in 32-byte halves.  The job of round k
//     runs right behind round k's barrier on ONE wavefront, the owner rotating from round to round (out of phase between the
//     workgroups of a CU), while the other wavefronts already produce round k+1; one workgroup barrier per round; a
//     chain's running sum travels from job to job through LDS.  (NS >= 8 -- 72+ chains -- packs jobs across round
//     boundaries, u = k * 9 NS + chain, three stage buffers; kept in the template, not launched.)
//   * what bounds it (measured): a dependent v_add_f32 of a lone wavefront costs 8.5 cycles (tools/ubench_chain.hip), so a
//     job is >= 544 cycles however few instructions surround the additions, and it sits on the round's critical path
//     together with its owner's own production (a build that runs every job twice is 28 us = 102 x 590 cycles slower).
//     Hence: everything that can be done by the producers in parallel is done there (the multiplications), the owner
//     keeps a raised priority until its next row is staged, and the job itself is only reads + adds.
//   Same arithmetic on the same texels in the same order as gn_match_kernel<.., EXACT>: identical bits.
//   Against round 2's producer / chain-wavefront form (gn_match_exact_batch_kernel: endpoints streamed, no texel cache):
//   66-69 vs 92 us on the 2048^2 headline batch, 141-143 vs 199 us on the 3-level batch, 156-162 vs 291 us on the 4096^2
//   pyramid, whose gathers miss the L2.
#pragma once
#include "gn_match.h"

namespace hsm {

constexpr int kXRow = 64 + kExactPad;  // floats per staged row: 16-byte aligned rows, chain lanes on distinct banks

#ifndef HSM_XEARLY  // round 6: wavefronts run ahead of the round barriers so that the owner's interval holds only the job --
                    // 2 = the balanced schedule (3 / 3 / 2 half-rows + job), 1 = the next owner produces two rows, 0 = round 3's schedule
#define HSM_XEARLY 2
#endif
#ifndef HSM_XBPC  // cached rows of the 17-row instantiation (see the kernel)
#define HSM_XBPC 15
#endif
#ifndef HSM_XBPC_MAIN  // ... of the four-producer form without a chain wavefront (the 4096-scan launch): the balanced schedule holds
#define HSM_XBPC_MAIN (HSM_XEARLY == 2 ? 13 : HSM_XBPC)  // a row's nine products across a barrier, which 15 cached rows do not leave room for
#endif
#ifndef HSM_XBPC_CW  // ... of the chain-wavefront form (five wavefronts per SIMD: 96 VGPRs)
#define HSM_XBPC_CW 6
#endif
#ifndef HSM_XLDS_AHEAD
#define HSM_XLDS_AHEAD 1
#endif
constexpr int kXRows = 9;  // staged rows per scan and round: the nine products of OccGridMapUtil.h:83-97
#ifndef HSM_XOWNER_PRIO_P  // priority the job's owner keeps while it produces its next row (until the next barrier)
#define HSM_XOWNER_PRIO_P 1
#endif
#ifndef HSM_XOWNER_SHIFT  // >= 0: the owner rotation of workgroup b starts at (b >> SHIFT) & 3 (workgroups of one CU out of phase)
#define HSM_XOWNER_SHIFT 8
#endif
#ifndef HSM_XPEEL  // the first GN step takes the endpoints from their load registers (gn_match_cached_kernel's peeled step)
#define HSM_XPEEL 1
#endif
#ifndef HSM_XEP_AHEAD  // endpoint loads in flight ahead of the beam being located in that step
#define HSM_XEP_AHEAD 2
#endif
#ifndef HSM_XGATHER_ALWAYS  // lane 0 re-reads its texel at every beam: exactly one load per beam, static waits, no branches
                            // (four instructions less per row on the owner's path: ~1 us per launch, profiles/r03/README.md)
#define HSM_XGATHER_ALWAYS 1
#endif
#ifndef HSM_XJOB_PRIO  // issue priority of a wavefront while it runs a chain job (the round's critical path)
#define HSM_XJOB_PRIO 3
#endif

// BPL rows of beams per lane, the first BPC of them with a cached texel (5 VGPRs per row; a chain job needs 16 VGPRs for
// the LDS rows it keeps in flight and 7 endpoint rows live in VGPRs, which 17 cached rows do not leave at 128);
// rows BPC .. BPL-1 gather in every step.
//
// CW (round 5): a CHAIN WAVEFRONT.  The workgroup gets one more wavefront (index NS) that produces nothing: it runs every
// round's chain job, the running sums never leave its registers inside a GN step, and no producer carries a job on top of
// its own row any more -- the round's critical path is max(job, production) instead of the owner's job + production.
// Launched for batches that leave a CU at most THREE such workgroups (hector_mi355.hip, launch_match_exact): the dispatcher
// places a workgroup only where every SIMD has room for ceil(5 / 4) = 2 of its wavefronts, so the third workgroup fits for
// certain only if a SIMD holds six wavefronts -- 80 VGPRs, i.e. fewer rows with a cached texel (BPC = HSM_XBPC_CW) -- and
// a fourth does not (tools/study/ubench_wg_placement.hip, profiles/r05/README.md 9); the launch bounds ask for five per SIMD
// (the compiler's figure includes the LDS: four workgroups), tests/test_kernel_resources.py holds the 80.
// (With at most TWO such workgroups per CU the second one fits at four wavefronts per SIMD as well -- 2/1/1/1 leaves two slots
// everywhere --, so launches of up to 2 x CUs workgroups take an instantiation with the full texel cache at 128 VGPRs: what a
// map that outgrows the L2s needs.)
// -DHSM_XTIMELINE (variant builds only: tools/study/exact_timeline.py): workgroup 0 stamps s_memtime at every round's barrier
// (arrival, release) and job end, per wavefront, into MatchParams::clock_probe [wave][row][4]; the last GN step's stamps stay
#ifdef HSM_XTIMELINE
#define HSM_XT(row, slot)                                                                                            \
  do {                                                                                                               \
    if (P.clock_probe != nullptr && blockIdx.x == 0 && lane == 0)                                                    \
      P.clock_probe[((size_t)wave * BPL + (size_t)(row)) * 4 + (slot)] = (unsigned long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define HSM_XT(row, slot) \
  do {                    \
  } while (0)
#endif

#ifndef HSM_XWGPRIO  // issue priority of a workgroup's producers by its dispatch order on the CU (blockIdx >> 8): the hardware arbitrates
#define HSM_XWGPRIO 3  // by priority, then AGE, so the workgroup dispatched last to a CU loses every tie and ends last (profiles/r06).
#endif                 // 0 = off; 1 = priority = order; 2 = (order + GN step) & 3: every workgroup is favoured in some steps; 3 = min(order, 2)
__device__ __forceinline__ void set_prio_uniform(int p) {  // s_setprio takes an immediate
  if (p <= 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}

// PROBE: the instantiation hsm_set_clock_probe switches a launch to (the four stamps cost this kernel 0.4 us -- four more live
// SGPRs -- so launches without a probe do not carry the code: profiles/r06/README.md 2)
template <int NS, int BPL, int BPC = BPL, bool CW = false, bool PROBE = false>
__global__ void __launch_bounds__(64 * (NS + (CW ? 1 : 0)), CW && 5 * BPC + 50 <= 96 ? 5 : 4) gn_match_exact_cached_kernel(const MatchParams P) {
  static_assert(BPC >= 1 && BPC <= BPL, "cached rows are a prefix of the rows");
  constexpr int NC = 9 * NS;  // chains per workgroup
  static_assert(!CW || NC <= 64, "the chain wavefront runs one job per round: lane = chain");
  // unit stride of a round.  A chain must not appear twice in one job (its rounds are sequential), so workgroups with
  // fewer than 64 chains pad the round to 64 units: one job per round, the lanes beyond NC idle.
  constexpr int NCP = NC >= 64 ? NC : 64;
  // stage buffers: a job of 64 units spans one round (NCP a multiple of 64) or two, and the next round is being
  // produced meanwhile
  constexpr int NB = NCP % 64 == 0 ? 2 : 3;
  // endpoint rows in VGPRs: what the LDS share of a workgroup (16 wavefronts per CU) does not hold next to the stage
  constexpr int kLdsShare = 160 * 1024 / (16 / NS);
  constexpr int kRowsFit = (kLdsShare - NB * NS * kXRows * kXRow * 4 - NC * 4 - 64) / (NS * 512);
  constexpr int RV = BPL <= kRowsFit ? 0 : BPL - kRowsFit;
  static_assert(RV < BPL && RV <= 8, "endpoint rows kept in VGPRs");
  static_assert(64 * (NS + (CW ? 1 : 0)) <= 1024, "one workgroup");
  // ONE shared object with the stage first: its LDS address must fit M0[15:0] (ds_write_addtid_b32 below)
  struct alignas(16) Smem {
    float stage[NB][NS][kXRows][kXRow];
    float runs[NC];
    int nmax;
    f2 pts[NS][BPL - RV][64];
  };
  static_assert(sizeof(Smem) <= kLdsShare, "16 wavefronts per CU");
  __shared__ Smem sm;
  auto& stage = sm.stage;
  auto& runs = sm.runs;
  int& nmax_s = sm.nmax;
  auto& lds_pts = sm.pts;
  static_assert(sizeof(sm.stage) < 65536, "stage rows are addressed through M0[15:0]");
  // a launch that takes part in the pose exchange (MatchParams::xp): the workgroups behind the matcher's own wait for an earlier
  // epoch and unpack it -- no registers, no LDS, no barrier of the matcher's are touched
  // (the four-producer forms only: the chain-wavefront forms sit exactly on the register budget their placement on a CU needs)
  const int match_blocks = (!CW && P.xp.world > 0) ? P.xp.match_blocks : (int)gridDim.x;
  if (!CW && (int)blockIdx.x >= match_blocks) {
    exchange_wait_unpack(P.xp, (int)blockIdx.x - match_blocks);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int slot = __builtin_amdgcn_readfirstlane(xcd_block((int)blockIdx.x, match_blocks, P.xcd_chunk) * NS + wave);
  // (MatchParams::perm: the batch in Morton order of its start poses -- neighbours in the launch are neighbours in the map; a slot
  // beyond the batch keeps an index beyond it)
  const int scan = (P.perm != nullptr && slot < P.batch) ? __builtin_amdgcn_readfirstlane(P.perm[slot]) : slot;
  const bool chain_wave = CW && wave == NS;   // wave-uniform
  const bool active = slot < P.batch && !chain_wave;  // inactive wavefronts (batch tail) run along with an empty scan: barriers, jobs
  int beg = 0, n = 0;
  float pw0 = 0.0f, pw1 = 0.0f, pw2 = 0.0f;
  if (active) {
    n = P.shared_n;
    if (P.offsets) {
      beg = P.offsets[scan];
      n = P.offsets[scan + 1] - beg;
    }
    pw0 = P.begin_world[3 * scan + 0];
    pw1 = P.begin_world[3 * scan + 1];
    pw2 = P.begin_world[3 * scan + 2];
  }
  // wave-uniform values live in SGPRs: this kernel has no VGPR to spare
  pw0 = uniform_f32(pw0), pw1 = uniform_f32(pw1), pw2 = uniform_f32(pw2);
  const float b0 = pw0, b1 = pw1, b2 = pw2;  // an empty scan passes its start estimate through (ScanMatcher.h:68,189)
  if (threadIdx.x == 0) nmax_s = 0;
  __syncthreads();
  if (lane == 0 && n > 0) atomicMax(&nmax_s, n);
  __syncthreads();
  const int nmax = nmax_s;  // workgroup-uniform
  if (nmax == 0) {          // nothing but empty scans
    if (!CW && active && P.xp.world > 0) exchange_post_pose(P.xp, scan, b0, b1, b2);
    if (active && lane == 0) {
      P.out_pose[3 * scan + 0] = b0;
      P.out_pose[3 * scan + 1] = b1;
      P.out_pose[3 * scan + 2] = b2;
    }
    return;
  }
  // rounds per GN step: the BPL cached rows (all of them: shorter scans pad with +-0 contributions), plus streamed
  // rounds for scans longer than the host's length hint
  const int rounds = BPL + (nmax > 64 * BPL ? (nmax - 64 * BPL + 63) >> 6 : 0);
  const int units = rounds * NCP;
  if (CW && chain_wave) {
    // every barrier of the producers' schedule, a job behind each round's: lane c = chain c, its row streams through two
    // 32-byte halves (see chain_job below), the running sum stays in a register until the step's last round
    __builtin_amdgcn_s_setprio(HSM_XJOB_PRIO);
    const int c = lane < NC ? lane : 0;
    for (int l = P.first_level; l >= P.last_level; --l) {
      const int gn_steps = __builtin_amdgcn_readfirstlane(P.lv[l].gn_steps);
      for (int it = 0; it < gn_steps; ++it) {
        float run = 0.0f;
        for (int k = 0; k < rounds; ++k) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          const f4v* pa = reinterpret_cast<const f4v*>(&stage[0][0][0][0] + ((k % NB) * NC + c) * kXRow);
          f4v a[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q] = pa[q];
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            const f4v p0 = a[2 * (h & 1)], p1 = a[2 * (h & 1) + 1];
            run += p0.x;
            run += p0.y;
            run += p0.z;
            run += p0.w;
            run += p1.x;
            run += p1.y;
            run += p1.z;
            run += p1.w;
            asm volatile("" : "+v"(run) : : "memory");
            if (h + 2 < 8) a[2 * (h & 1)] = pa[2 * h + 4], a[2 * (h & 1) + 1] = pa[2 * h + 5];
            asm volatile("" ::: "memory");
          }
        }
        if (lane < NC) runs[lane] = run;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the totals are published
      }
    }
    return;
  }
  // hsm_set_clock_probe: workgroup 0 stamps the shader-clock counter and the 100 MHz wall clock when it starts and when it ends
  // (both are scalar reads; the start stamps wait in SGPRs and go out with the end stamps, where the texel cache no longer holds
  // the VGPRs)
  const unsigned long long probe_t0 = PROBE ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
  const unsigned long long probe_w0 = PROBE ? wall_clock64() : 0ull;
#ifdef HSM_XTIMELINE_WG  // (variant builds: start / end wall clock (100 MHz) and XCC id of every workgroup, [block][4] behind the other stamps)
  if (P.clock_probe != nullptr && wave == 0 && lane == 0) {
    P.clock_probe[1024 + 4 * (size_t)blockIdx.x + 0] = wall_clock64();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    P.clock_probe[1024 + 4 * (size_t)blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
  const float2* __restrict__ pts = P.pts + (n > 0 ? beg : 0);  // an empty scan's loads (clamped to element 0) stay inside the array
  f2(*mine)[64] = lds_pts[wave];
  f2 pv[RV > 0 ? RV : 1];
  // Endpoint staging.  All wavefronts of a launch start together and each needs its 8.6 KB of endpoints: 35 MB at once,
  // 6 us with nothing to compute.  As in gn_match_cached_kernel the FIRST GN step of the first level is peeled: beam k
  // takes its endpoint from the register of a load issued kEpAhead beams earlier (and stores it, scaled for the level,
  // for the later steps), every lane gathers its texel (a level's first step), and all waits are counted from
  // peel_schedule()'s static issue order.
  constexpr bool kPeel = HSM_XPEEL != 0;
  constexpr int kEpAhead = HSM_XEP_AHEAD < BPL ? HSM_XEP_AHEAD : BPL - 1;
  static_assert(BPL <= 31, "PeelSchedule holds 32 positions per load kind");
  constexpr PeelSchedule kSched = peel_schedule(BPL, kEpAhead);
  const bool peel = kPeel && P.lv[P.first_level].gn_steps > 0;  // workgroup-uniform
  if (!peel) {
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
      const int i = lane + 64 * k;
      const float2 q = i < n ? pts[i] : make_float2(1.0e30f, 1.0e30f);  // padding: exact +-0 contributions (gn_match_kernel)
      if (k < RV)
        pv[k] = f2{q.x, q.y};
      else
        mine[k - RV][lane] = f2{q.x, q.y};
    }
  }
  f4v tq[BPC];
  unsigned toff[BPC];
  Acc9 acc;
  acc.zero();
  float reg_scale = 1.0f;
  int step_no = 0;
  const int owner_phase = HSM_XOWNER_SHIFT >= 0 ? (int)((blockIdx.x >> (HSM_XOWNER_SHIFT >= 0 ? HSM_XOWNER_SHIFT : 0)) & 3u) : 0;
  // LDS byte address of this wavefront's stage rows in buffer 0 (wave-uniform; a generic LDS pointer's low half)
  const unsigned st_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&stage[0][wave][0][0]);
  for (int l = P.first_level; l >= P.last_level; --l) {
    const LevelView& L = P.lv[l];
    float ex, ey, eth;
    affine_apply(L.mapTworld, pw0, pw1, ex, ey);
    ex = uniform_f32(ex), ey = uniform_f32(ey);
    eth = pw2;
    const float ps = L.pt_scale;
    const int gn_steps = L.gn_steps;
    const LevelRegs R = level_regs<kLayoutQuad>(L);
    const float ratio = ps / reg_scale;  // powers of two: exact
    reg_scale = ps;
    const bool peel_here = peel && l == P.first_level;  // the peeled step stages the endpoints, scaled for this level
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
      if (!peel_here && ratio != 1.0f) {
        if (k < RV)
          pv[k] *= f2{ratio, ratio};
        else
          mine[k - RV][lane] *= f2{ratio, ratio};
      }
      if (k < BPC) toff[k] = 0xffffffffu;  // never a texel offset: every beam gathers in the level's first step
    }
    unsigned zero_off = (unsigned)R.zero_index << 4;
    asm volatile("" : "+v"(zero_off));
    // one GN step; FIRST = the peeled step (compile-time)
    auto gn_step = [&](auto FIRST) {
      constexpr bool kFirst = decltype(FIRST)::value;
#ifdef HSM_XTIMELINE_STEPS  // (variant builds: one stamp per GN step of workgroup 0's first wavefront, behind the per-round stamps' block)
      if (P.clock_probe != nullptr && blockIdx.x == 0 && wave == 0 && lane == 0 && step_no < 32)
        P.clock_probe[(size_t)4 * 31 * 4 + step_no] = (unsigned long long)__builtin_readcyclecounter();
#endif
      f2 pq[kFirst ? BPL : 1];  // the peeled step's endpoint load registers
      auto endpoint_issue = [&](int k) {
        int i = max(min((int)lane_id_now() + 64 * k, n - 1), 0);
        asm volatile("" : "+v"(i));  // the offset is computed at the load, not hoisted into 17 VGPRs
        const unsigned byte_off = (unsigned)i << 3;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(pq[kFirst ? k : 0]) : "v"(byte_off), "s"(pts) : "memory");
      };
      if (kFirst) {
#pragma unroll
        for (int k = 0; k <= kEpAhead; ++k) endpoint_issue(k);
      }
      float sinRot, cosRot;
      sincos_f32<true>(eth, sinRot, cosRot);
      const f2 o2 = step_origin(ex, ey);
      const f2 e2 = f2{uniform_f32(o2.x), uniform_f32(o2.y)};
      const f2 cs = f2{uniform_f32(cosRot), uniform_f32(sinRot)}, sc = f2{cs.y, cs.x};
      auto endpoint = [&](int k) -> f2 {
        if (kFirst) {  // from its load register (beam k's gather is the next load in issue order), padded, scaled, kept
          wait_vmcnt(kSched.posG[k] - kSched.posE[k] - 1, pq[kFirst ? k : 0]);
          const bool pad = (int)lane_id_now() + 64 * k >= n;
          const f2 q = pq[kFirst ? k : 0];
          const f2 p = f2{(pad ? 1.0e30f : q.x) * ps, (pad ? 1.0e30f : q.y) * ps};
          if (k < RV)
            pv[k < RV ? k : 0] = p;
          else
            mine[k < RV ? 0 : k - RV][lane] = p;
          return p;
        }
        return k < RV ? pv[k < RV ? k : 0] : mine[k < RV ? 0 : k - RV][lane];
      };
      // rotate, bounds test, cell offset, fractions; gather only in the lanes whose cell changed (gn_match_cached_kernel)
      f4v tu[2];  // texels of the uncached rows (k >= BPC), alternating
      auto locate = [&](int k, f2 p, BeamRot& r, float& fx, float& fy) -> unsigned long long {
        r.r.x = cs.x * p.x - sc.x * p.y;
        r.r.y = cs.y * p.x + sc.y * p.y;
        const CellCoord q = cell_coord(R, f2{e2.x + r.r.x, e2.y + r.r.y});
        fx = q.fx;
        fy = q.fy;
        unsigned idx = quad_index(q.ix, q.iy, R.tiles_x, R.sx);
        asm volatile("" : "+v"(idx));
        const unsigned off = q.oob ? zero_off : idx << 4;
        if (k >= BPC) {  // uncached row: every lane gathers
          asm volatile("global_load_dwordx4 %[t], %[o], %[b]" : [t] "=v"(tu[k & 1]) : [o] "v"(off), [b] "s"(R.quad) : "memory");
          return ~0ull;
        }
        const int kc = k < BPC ? k : 0;
        if (kFirst) {  // a level's first step: every lane gathers (toff[] holds no offset yet); one load, statically counted
          asm volatile("global_load_dwordx4 %[t], %[o], %[b]" : [t] "=v"(tq[kc]) : [o] "v"(off), [b] "s"(R.quad) : "memory");
          toff[kc] = off;
          return ~0ull;
        }
        unsigned long long moved, saved;
#if HSM_XGATHER_ALWAYS
        asm volatile(
            "v_cmp_ne_u32 vcc, %[o], %[to]\n\t"
            "s_or_b32 vcc_lo, vcc_lo, 1\n\t"
            "s_and_saveexec_b64 %[sv], vcc\n\t"
            "global_load_dwordx4 %[t], %[o], %[b]\n\t"
            "v_mov_b32 %[to], %[o]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [t] "+v"(tq[kc]), [to] "+v"(toff[kc]), [sv] "=&s"(saved)
            : [o] "v"(off), [b] "s"(R.quad)
            : "vcc", "scc", "memory");
        return ~0ull;
#endif
        asm volatile(
            "v_cmp_ne_u32 vcc, %[o], %[to]\n\t"
            "s_mov_b64 %[mv], vcc\n\t"
            "s_and_saveexec_b64 %[sv], vcc\n\t"
            "s_cbranch_execz 1f\n\t"
            "global_load_dwordx4 %[t], %[o], %[b]\n\t"
            "v_mov_b32 %[to], %[o]\n\t"
            "1:\n\t"
            "s_mov_b64 exec, %[sv]"
            : [t] "+v"(tq[kc]), [to] "+v"(toff[kc]), [sv] "=&s"(saved), [mv] "=&s"(moved)
            : [o] "v"(off), [b] "s"(R.quad)
            : "vcc", "scc", "memory");
        return moved;
      };
      auto texel_ready = [&](int k, unsigned long long next_moved, bool has_next) {
        f4v& tx = k < BPC ? tq[k < BPC ? k : 0] : tu[k & 1];
        if (kFirst) {  // static schedule: everything issued after beam k's gather may still be in flight
          wait_vmcnt((has_next ? kSched.posG[k + 1] + 1 : kSched.total) - kSched.posG[k] - 1, tx);
        } else if (has_next && (k + 1 >= BPC || HSM_XGATHER_ALWAYS)) {  // the next row's gather is unconditional
          asm volatile("s_waitcnt vmcnt(1)" : "+v"(tx) : : "memory");
        } else if (has_next) {
          asm volatile(
              "s_cmp_eq_u64 %[m], 0\n\t"
              "s_cbranch_scc1 1f\n\t"
              "s_waitcnt vmcnt(1)\n\t"
              "s_branch 2f\n\t"
              "1:\n\t"
              "s_waitcnt vmcnt(0)\n\t"
              "2:"
              : "+v"(tx)
              : [m] "s"(next_moved)
              : "scc", "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(tx) : : "memory");
        }
      };
      // the four staged values of one beam (OccGridMapUtil.h:332-346 and :80-87), with the source's signs:
      //   gx = -((P00-P10)*xFacInv + (P01-P11)*fx) == (P10-P00)*xFacInv + (P11-P01)*fx   (negation commutes with rounding)
      //   rotDeriv = (-ry)*gx + rx*gy == rx*gy - ry*gx,  (rx, ry) = R(theta) p shared with the transform (gn_match.h)
      // the nine products of :83-97, one rounding each like the reference's; the chain lane only adds
      // (in two halves: the four terms of a beam -- gx, gy, funVal, rotDeriv -- are what a wavefront that runs ahead of a barrier keeps
      // in registers; the nine products are formed from them when the row is staged)
      auto beam_terms4 = [&](float i0, float i1, float i2, float i3, const BeamRot& r, float fx, float fy, float (&tm)[4]) {
        const float xFacInv = 1.0f - fx, yFacInv = 1.0f - fy;
        const float M = ((i0 * xFacInv + i1 * fx) * (yFacInv)) + ((i2 * xFacInv + i3 * fx) * (fy));
        tm[0] = ((i1 - i0) * xFacInv) + ((i3 - i2) * fx);  // gx
        tm[1] = ((i2 - i0) * yFacInv) + ((i3 - i1) * fy);  // gy
        tm[2] = 1.0f - M;                                  // funVal
        tm[3] = r.r.x * tm[1] - r.r.y * tm[0];             // rotDeriv
      };
      auto products_of = [&](const float (&tm)[4], float (&pp)[9]) {
        const float gx = tm[0], gy = tm[1], funVal = tm[2], rotDeriv = tm[3];
        pp[0] = gx * funVal, pp[1] = gy * funVal, pp[2] = rotDeriv * funVal;
        pp[3] = gx * gx, pp[4] = gy * gy, pp[5] = rotDeriv * rotDeriv;
        pp[6] = gx * gy, pp[7] = gx * rotDeriv, pp[8] = gy * rotDeriv;
      };
      auto products = [&](float i0, float i1, float i2, float i3, const BeamRot& r, float fx, float fy, float (&pp)[9]) {
        float tm[4];
        beam_terms4(i0, i1, i2, i3, r, fx, fy, tm);
        products_of(tm, pp);
      };
      // lane l at row + 4 l: ds_write_addtid_b32 (address = M0 + offset + 4 * lane) needs no address VGPR and half the LDS
      // cycles of ds_write_b32 (MI355X_MICROARCH.md, LDS).  M0 is reserved, not allocatable: nothing else in this kernel
      // uses it (gfx9+ DS operations do not)
      auto stage_write = [&](int k, const float (&pp)[9]) {
        const int buf = k % NB;
        const unsigned m0v = st_wave + (unsigned)buf * (NS * kXRows * kXRow * 4);
        asm volatile(
            "s_mov_b32 m0, %[m]\n\t"
            "s_nop 0\n\t"
            "ds_write_addtid_b32 %[p0] offset:%[o0]\n\t"
            "ds_write_addtid_b32 %[p1] offset:%[o1]\n\t"
            "ds_write_addtid_b32 %[p2] offset:%[o2]\n\t"
            "ds_write_addtid_b32 %[p3] offset:%[o3]\n\t"
            "ds_write_addtid_b32 %[p4] offset:%[o4]\n\t"
            "ds_write_addtid_b32 %[p5] offset:%[o5]\n\t"
            "ds_write_addtid_b32 %[p6] offset:%[o6]\n\t"
            "ds_write_addtid_b32 %[p7] offset:%[o7]\n\t"
            "ds_write_addtid_b32 %[p8] offset:%[o8]"
            :
            : [m] "s"(m0v), [p0] "v"(pp[0]), [p1] "v"(pp[1]), [p2] "v"(pp[2]), [p3] "v"(pp[3]), [p4] "v"(pp[4]), [p5] "v"(pp[5]), [p6] "v"(pp[6]),
              [p7] "v"(pp[7]), [p8] "v"(pp[8]), [o0] "n"(0 * kXRow * 4), [o1] "n"(1 * kXRow * 4), [o2] "n"(2 * kXRow * 4),
              [o3] "n"(3 * kXRow * 4), [o4] "n"(4 * kXRow * 4), [o5] "n"(5 * kXRow * 4), [o6] "n"(6 * kXRow * 4),
              [o7] "n"(7 * kXRow * 4), [o8] "n"(8 * kXRow * 4)
            : "memory");
      };
      // the four staged values of one beam (OccGridMapUtil.h:332-346 and :80-87), with the source's signs:
      //   gx = -((P00-P10)*xFacInv + (P01-P11)*fx) == (P10-P00)*xFacInv + (P11-P01)*fx   (negation commutes with rounding)
      //   rotDeriv = (-ry)*gx + rx*gy == rx*gy - ry*gx,  (rx, ry) = R(theta) p shared with the transform (gn_match.h)
      auto produce = [&](int k, float i0, float i1, float i2, float i3, const BeamRot& r, float fx, float fy) {
        float pp[9];
        products(i0, i1, i2, i3, r, fx, fy, pp);
        stage_write(k, pp);
      };
      // one chain job: lane l runs unit u = 64 j + l = (round ku, chain c): 64 dependent additions on top of the chain's
      // running sum.  The chain's row streams through two 32-byte halves (16 VGPRs): a half is refilled right after its
      // values are consumed, one s_waitcnt per eight additions.  (Measured, profiles/r03/README.md: a dependent v_add_f32 of
      // a lone wavefront costs 8.5 cycles, so the 64 additions are the job's length and the reads hide in their bubbles.)
      auto chain_job = [&](int j, int k) {
        // one job per round (NCP == 64): job j is round k, lane l is chain l -- nothing to derive
        constexpr bool kOneJob = NCP == 64;
        // otherwise the lane index is re-read here (volatile asm): everything below depends on it, so the per-lane unit /
        // address arithmetic of the ~20 jobs of a step is not hoisted out of the GN loop into VGPRs that are not there
        const int u = kOneJob ? 64 * k + lane : 64 * j + lane_id_now();
        // a job that completes with round k holds units of rounds k-1 and k only (64 <= NCP): no division
        const bool prev = kOneJob ? false : u < k * NCP;
        const int ku = prev ? k - 1 : k, c = kOneJob ? lane : u - ku * NCP;  // round, chain
        if ((kOneJob || u < units) && c < NC) {
          const int buf = prev ? (k + NB - 1) % NB : k % NB;
          float run = ku == 0 ? 0.0f : runs[c];
          // chain c = 9 scan + term is row c of the buffer
          const f4v* pa = reinterpret_cast<const f4v*>(&stage[0][0][0][0] + (buf * NC + c) * kXRow);
          f4v a[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q] = pa[q];
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            const f4v p0 = a[2 * (h & 1)], p1 = a[2 * (h & 1) + 1];
            run += p0.x;
            run += p0.y;
            run += p0.z;
            run += p0.w;
            run += p1.x;
            run += p1.y;
            run += p1.z;
            run += p1.w;
            asm volatile("" : "+v"(run) : : "memory");
            if (h + 2 < 8) a[2 * (h & 1)] = pa[2 * h + 4], a[2 * (h & 1) + 1] = pa[2 * h + 5];
            asm volatile("" ::: "memory");
          }
          runs[c] = run;
        }
      };
      const int wg_order = (int)((blockIdx.x >> 8) & 3u);
      const int wg_prio = HSM_XWGPRIO == 1 ? wg_order : HSM_XWGPRIO == 2 ? ((wg_order + step_no) & 3) : HSM_XWGPRIO == 3 ? min(wg_order, 2) : 0;
      if (HSM_XWGPRIO != 0 && !CW) set_prio_uniform(wg_prio);
      // round k is staged: meet, then (one wavefront) run the chain jobs that are complete with it
      const int my_rounds =
          __builtin_amdgcn_readfirstlane((int)((unsigned)(wave + 64 * NS - (step_no + owner_phase) % NS) % (unsigned)NS));
      auto round_done = [&](int k, bool last_round) {
        if (HSM_XOWNER_PRIO_P != 0) __builtin_amdgcn_s_setprio(0);
        if (k < BPL) HSM_XT(k, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (k < BPL) HSM_XT(k, 1);
        if (CW) return;  // the chain wavefront runs the job
        const int j_lo = (k * NCP) >> 6;
        const int j_hi = last_round ? (units + 63) >> 6 : ((k + 1) * NCP) >> 6;
        if (j_lo >= j_hi) return;
        // round k's owner is wave (k + step_no + phase) % NS.  Compared afresh in an SGPR at every round: kept as booleans
        // across the unrolled rounds the NS outcomes come back through v_cndmask / v_cmp pairs in every round
        int mine_now = my_rounds;
        asm volatile("" : "+s"(mine_now));
        if ((int)((unsigned)k % (unsigned)NS) != mine_now) return;
        __builtin_amdgcn_s_setprio(HSM_XJOB_PRIO);
#if defined(HSM_EXPERIMENTS) && defined(HSM_XWHATIF) && HSM_XWHATIF == 1  // timing experiment: every job runs twice (same sums: the second run starts from the first's input)
        for (int j = j_lo; j < j_hi; ++j) {
          const float keep = lane < NC ? runs[lane] : 0.0f;
          chain_job(j, k);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane < NC) runs[lane] = keep;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#endif
        for (int j = j_lo; j < j_hi; ++j) chain_job(j, k);
        __builtin_amdgcn_s_setprio(HSM_XOWNER_PRIO_P);
        if (k < BPL) HSM_XT(k, 2);
      };
      {
        BeamRot rc, rn;
        float fxc, fyc, fxn = 0.0f, fyn = 0.0f;
        unsigned long long next_moved = 0ull;
        constexpr bool kAhead = HSM_XLDS_AHEAD != 0 && !kFirst;
        // (the first, peeled step keeps the rotating-owner schedule: measured slower with either early form -- all lanes gather, the
        // endpoints stream from HBM, its waits are counted from a static issue order)
#ifndef HSM_XEARLY_FIRST
#define HSM_XEARLY_FIRST 0
#endif
        // (... and so does the 17-row form with fifteen cached rows, which has no registers left to hold a row across a barrier: the
        // instantiation for maps that outgrow the L2s -- 4096^2 pyramid: 145 us with 15 cached rows and the rotating owner, 153 us
        // with 13 and the balanced schedule, profiles/r06)
        constexpr bool kBalanced = HSM_XEARLY == 2 && !CW && NCP == 64 && (!kFirst || HSM_XEARLY_FIRST != 0) && (BPL < 17 || BPC <= HSM_XBPC_MAIN);
        constexpr bool kEarly = HSM_XEARLY == 1 && !CW && NCP == 64 && !kFirst;
        f2 p_next = f2{0.0f, 0.0f};
        if (kAhead) p_next = endpoint(BPL > 1 ? 1 : 0);
        locate(0, endpoint(0), rc, fxc, fyc);
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
          if (kFirst && k + 1 + kEpAhead < BPL) endpoint_issue(k + 1 + kEpAhead);
          if (kAhead) {  // endpoint of beam k+2 read from LDS before beam k+1 is located (two more VGPRs)
            const f2 p_cur = p_next;
            if (k + 2 < BPL) p_next = endpoint(k + 2);
            if (k + 1 < BPL) next_moved = locate(k + 1, p_cur, rn, fxn, fyn);
          } else {
            if (k + 1 < BPL) next_moved = locate(k + 1, endpoint(k + 1), rn, fxn, fyn);
          }
#if defined(HSM_EXPERIMENTS) && defined(HSM_XWHATIF) && HSM_XWHATIF == 2  // timing experiment: a second workgroup barrier per round
          asm volatile("s_barrier" ::: "memory");
#endif
          if (kBalanced) {
            // Round 6.  Where a round's time went (profiles/r06/README.md: s_memtime per round): the owner of round k's job ran it behind
            // barrier k and THEN produced its row k+1 -- job (~700 cycles) + production (~500) on every round's critical path, three
            // wavefronts parked at barrier k+1 meanwhile (SQ_WAIT_ANY 0.53 of the wavefront cycles).  Now a wavefront runs AHEAD of
            // the barriers by up to one row -- the products of a row that is early wait in nine registers until the barrier that
            // frees its stage buffer -- on a schedule that gives the owner's interval nothing but the job and spreads its four
            // productions over the other three: with o = (k - own round) mod 4, a = locate row k+1, b = products of row k,
            //   o = 3: a | barrier k-1 | b, stage          o = 0: a, b (held) | barrier k-1 | stage
            //   o = 1: a, b (held) | barrier k-1 | stage, JOB k-1 | barrier k          o = 2: a, b, stage (no barrier)
            // i.e. intervals of 3, 3, 2 half-rows and the job.  Every wavefront still meets each round's barrier exactly once, in
            // order; same products, same order of the additions: identical bits.
            int mine_now = my_rounds;
            asm volatile("" : "+s"(mine_now));  // (compared afresh in an SGPR at every round: see round_done)
            const int o = (int)((unsigned)(k + NS - mine_now) % (unsigned)NS);
            if (o == 3 && k >= 1) {
              if (HSM_XWGPRIO == 0 && HSM_XOWNER_PRIO_P != 0) __builtin_amdgcn_s_setprio(0);
              asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            texel_ready(k, next_moved, k + 1 < BPL);
            const f4v& tx = k < BPC ? tq[k < BPC ? k : 0] : tu[k & 1];
            float pp[9];
            products(tx.x, tx.y, tx.z, tx.w, rc, fxc, fyc, pp);
            if (o >= 2 || k == 0) {
              stage_write(k, pp);
              if (k == 0 && o == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (no job -1 to run)
            } else {  // o = 0, 1 with k >= 1: the products wait for barrier k-1
              if (HSM_XWGPRIO == 0 && HSM_XOWNER_PRIO_P != 0) __builtin_amdgcn_s_setprio(0);
              asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
              stage_write(k, pp);
              if (o == 1) {
                __builtin_amdgcn_s_setprio(HSM_XJOB_PRIO);
                chain_job(k - 1, k - 1);
                if (HSM_XWGPRIO != 0) set_prio_uniform(wg_prio); else __builtin_amdgcn_s_setprio(0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
              }
            }
            if (k == BPL - 1 && o != 1) {  // the barrier of the last cached row (o = 1 has met it), and its job
              asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
              if (o == 0) {
                __builtin_amdgcn_s_setprio(HSM_XJOB_PRIO);
                chain_job(k, k);
                if (HSM_XWGPRIO != 0) set_prio_uniform(wg_prio); else __builtin_amdgcn_s_setprio(0);
              }
            }
          } else if (kEarly) {
            texel_ready(k, next_moved, k + 1 < BPL);
            // Round 6.  The owner of round k's job used to run it behind barrier k and THEN produce its row k+1 -- job + production
            // on every round's critical path, three wavefronts parked at barrier k+1 meanwhile (SQ_WAIT_ANY 0.53 of the wavefront
            // cycles, profiles/r06/stall_table.txt).  Now the owner-to-be produces row k+1 BEFORE barrier k (its products wait in nine
            // registers: stage buffer (k+1) % 2 is still being read by job k-1), so behind barrier k it only stages them and runs
            // the job: an interval lasts max(job, two productions) instead of job + production.  Every wavefront still meets
            // exactly one barrier per round; same products, same order of the additions.
            const f4v& tx = k < BPC ? tq[k < BPC ? k : 0] : tu[k & 1];
            float pp[9];
            products(tx.x, tx.y, tx.z, tx.w, rc, fxc, fyc, pp);
            int mine_now = my_rounds;
            asm volatile("" : "+s"(mine_now));  // (compared afresh in an SGPR at every round: see round_done)
            const bool own_k = (int)((unsigned)k % (unsigned)NS) == mine_now;
            const bool own_km1 = k >= 1 && (int)((unsigned)(k + NS - 1) % (unsigned)NS) == mine_now;
            if (own_km1) {  // row k was produced early: barrier k-1, stage it, job k-1, barrier k
              if (HSM_XOWNER_PRIO_P != 0) __builtin_amdgcn_s_setprio(0);
              HSM_XT(k - 1, 0);
              asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
              HSM_XT(k - 1, 1);
              stage_write(k, pp);
              __builtin_amdgcn_s_setprio(HSM_XJOB_PRIO);
              chain_job(k - 1, k - 1);
              __builtin_amdgcn_s_setprio(0);
              HSM_XT(k - 1, 2);
              HSM_XT(k, 0);
              asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
              HSM_XT(k, 1);
            } else {
              stage_write(k, pp);
              if (own_k && k + 1 < BPL) {
                __builtin_amdgcn_s_setprio(HSM_XOWNER_PRIO_P);  // no barrier here: this wavefront produces row k+1 first
              } else {
                HSM_XT(k, 0);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                HSM_XT(k, 1);
                if (own_k) {  // the last cached row is this wavefront's: nothing to produce ahead
                  __builtin_amdgcn_s_setprio(HSM_XJOB_PRIO);
                  chain_job(k, k);
                  __builtin_amdgcn_s_setprio(0);
                  HSM_XT(k, 2);
                }
              }
            }
          } else {
            texel_ready(k, next_moved, k + 1 < BPL);
            {
              const f4v& tx = k < BPC ? tq[k < BPC ? k : 0] : tu[k & 1];
              produce(k, tx.x, tx.y, tx.z, tx.w, rc, fxc, fyc);
            }
            round_done(k, k == BPL - 1 && rounds == BPL);
          }
          rc = rn;
          fxc = fxn;
          fyc = fyn;
        }
      }
      // scans longer than the 64 * BPL cached beams: the rest streams from memory, uncached
      for (int k = BPL; k < rounds; ++k) {
        const int i = 64 * k + lane;
        const float2 p = i < n ? pts[i] : make_float2(1.0e30f, 1.0e30f);
        BeamRot r;
        const BeamSample b = beam_fetch<kLayoutQuad>(R, e2, cs, sc, f2{p.x * ps, p.y * ps}, r);
        produce(k, b.lo.x, b.lo.y, b.hi.x, b.hi.y, r, b.X.y, b.Y.y);
        round_done(k, k == rounds - 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the last jobs have published the totals
      {
        const float* tt = &runs[9 * wave];
        acc.d01 = f2{uniform_f32(tt[0]), uniform_f32(tt[1])};
        acc.d2 = uniform_f32(tt[2]);
        acc.hd = f2{uniform_f32(tt[3]), uniform_f32(tt[4])};
        acc.h22 = uniform_f32(tt[5]);
        acc.h01 = uniform_f32(tt[6]);
        acc.hr = f2{uniform_f32(tt[7]), uniform_f32(tt[8])};
      }
      gn_solve_and_step(acc, ex, ey, eth);
      ex = uniform_f32(ex), ey = uniform_f32(ey), eth = uniform_f32(eth);
    };
    int it = 0;
    if (kPeel && peel_here) {
      gn_step(std::true_type{});
      ++step_no;
      it = 1;
    }
    for (; it < gn_steps; ++it, ++step_no) gn_step(std::false_type{});
    eth = normalize_angle<true>(eth);
    affine_apply(L.worldTmap, ex, ey, pw0, pw1);
    pw0 = uniform_f32(pw0), pw1 = uniform_f32(pw1), pw2 = uniform_f32(eth);
  }
#if !defined(HSM_XTIMELINE) && !defined(HSM_XTIMELINE_STEPS) && !defined(HSM_XTIMELINE_WG)
  if (PROBE && P.clock_probe != nullptr && blockIdx.x == 0 && wave == 0 && lane == 0) {
    P.clock_probe[0] = probe_t0, P.clock_probe[1] = probe_w0;
    P.clock_probe[2] = (unsigned long long)__builtin_readcyclecounter();
    P.clock_probe[3] = wall_clock64();
  }
#endif
#ifdef HSM_XTIMELINE_STEPS
  if (P.clock_probe != nullptr && blockIdx.x == 0 && wave == 0 && lane == 0 && step_no < 32)
    P.clock_probe[(size_t)4 * 31 * 4 + step_no] = (unsigned long long)__builtin_readcyclecounter();
#endif
#ifdef HSM_XTIMELINE_WG
  if (P.clock_probe != nullptr && wave == 0 && lane == 0) P.clock_probe[1024 + 4 * (size_t)blockIdx.x + 1] = wall_clock64();
#endif
  if (!CW && active && P.xp.world > 0) exchange_post_pose(P.xp, scan, n == 0 ? b0 : pw0, n == 0 ? b1 : pw1, n == 0 ? b2 : pw2);
  if (active && lane == 0) {
    const bool empty = n == 0;
    P.out_pose[3 * scan + 0] = empty ? b0 : pw0;
    P.out_pose[3 * scan + 1] = empty ? b1 : pw1;
    P.out_pose[3 * scan + 2] = empty ? b2 : pw2;
    if (P.out_cov && !empty) {
      float* c = P.out_cov + 9 * scan;
      c[0] = acc.hd.x; c[1] = acc.h01; c[2] = acc.hr.x;
      c[3] = acc.h01; c[4] = acc.hd.y; c[5] = acc.hr.y;
      c[6] = acc.hr.x; c[7] = acc.hr.y; c[8] = acc.h22;
    }
  }
}

}  // namespace hsm
