#!/usr/bin/env python
"""Per-wave start/end time stamps of headline launches (library built with -DHSM_EXPERIMENTS -DHSM_EXP_TIMESTAMPS: the kernel writes them
over the covariance output): how do finish times spread over XCDs / CUs / SIMDs, how long does the endpoint staging
take, and is a wave's lifetime a property of its scan (data) or of where it ran (hardware)?
usage: HSM_LIB=.../libhector_mi355_ts.so python tools/exp_wave_timeline.py [--shuffle]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def cu_finish(key, e):
    ks = np.unique(key)
    return np.array([e[key == k].max() for k in ks]) if len(ks) == 256 else np.zeros(256)


def analyse(u, B, prev):
    t0 = u[:, 0].astype(np.uint64) | (u[:, 1].astype(np.uint64) << np.uint64(32))
    t1 = u[:, 2].astype(np.uint64) | (u[:, 3].astype(np.uint64) << np.uint64(32))
    base = t0.min()
    b = (t0 - base).astype(np.float64) / 100.0  # 100 MHz -> us
    e = (t1 - base).astype(np.float64) / 100.0
    hwid, xcc, blk = u[:, 4], u[:, 5] & 0xF, u[:, 6]
    stage = u[:, 7].astype(np.float64) / 100.0  # kernel entry -> endpoints staged (peeled form: -> first step begins)
    entry = b - stage
    print(f"kernel entry -> first GN step: min/median/p90/max {stage.min():.2f}/{np.median(stage):.2f}/"
          f"{np.percentile(stage, 90):.2f}/{stage.max():.2f} us; first entry {entry.min():.2f} us, last entry {entry.max():.2f} us "
          f"(relative to the first wave's first step)")
    sclk = u[:, 8].astype(np.float64) / np.maximum(e - b, 1e-3)  # shader-clock cycles per us of the 100 MHz wall clock = MHz
    print(f"shader clock over the waves' lifetimes (s_memtime / wall clock): min/median/max {sclk.min():.0f}/{np.median(sclk):.0f}/{sclk.max():.0f} MHz")
    cu = (hwid >> 8) & 0xF
    se = (hwid >> 13) & 0x7
    print(f"waves {B}: start min/median/max {b.min():.1f}/{np.median(b):.1f}/{b.max():.1f} us; end min/median/p90/max "
          f"{e.min():.1f}/{np.median(e):.1f}/{np.percentile(e, 90):.1f}/{e.max():.1f} us; lifetime median {np.median(e - b):.1f}")
    for x in range(8):
        sel = xcc == x
        if sel.any():
            print(f"  XCC {x}: waves {sel.sum():4d} start max {b[sel].max():5.1f} end median {np.median(e[sel]):5.1f} max {e[sel].max():5.1f}  blocks {blk[sel].min()}..{blk[sel].max()} (mod 8 = {sorted(set((blk[sel] % 8).tolist()))})")
    # per CU finish (xcc, se, cu)
    key = xcc.astype(np.int64) * 1000 + se.astype(np.int64) * 100 + cu.astype(np.int64)
    fin = {k: e[key == k].max() for k in np.unique(key)}
    f = np.array(list(fin.values()))
    print(f"  per-CU finish: {len(f)} CUs, min {f.min():.1f} median {np.median(f):.1f} p90 {np.percentile(f, 90):.1f} max {f.max():.1f} us; "
          f"mean idle before the launch ends {np.mean(f.max() - f):.1f} us")
    # per SIMD: the four co-resident waves' finish times in order (HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13)
    simd = (hwid >> 4) & 0x3
    skey = key * 10 + simd.astype(np.int64)
    order = []
    for k in np.unique(skey):
        ee = np.sort(e[skey == k])
        if len(ee) == 4:
            order.append(ee)
    if order:
        o = np.array(order)
        print(f"  per-SIMD finish order ({len(o)} SIMDs with 4 waves): mean 1st..4th {o.mean(0).round(1).tolist()} us; "
              f"SIMD busy-slot fraction {float((o.sum(1) / (4 * e.max())).mean()):.3f}")
    life = e - b
    print("  lifetime by batch sixteenth (scan index):", [round(float(life[i * B // 16:(i + 1) * B // 16].mean()), 1) for i in range(16)])
    print("  lifetime by XCC:", [round(float(life[xcc == x].mean()), 1) for x in range(8)])
    if prev is not None:
        p_life, p_key, p_e = prev
        same_place = float((p_key == key).mean())
        print(f"  two launches of the same batch: correlation of per-scan lifetime {float(np.corrcoef(life, p_life)[0, 1]):.3f}, "
              f"of per-CU finish time {float(np.corrcoef(cu_finish(key, e), cu_finish(p_key, p_e))[0, 1]):.3f}; "
              f"{same_place:.2f} of the scans ran on the same CU")
    hist, edges = np.histogram(e, bins=12)
    print("  end-time histogram:", [(round(float(edges[i]), 1), int(hist[i])) for i in range(len(hist))])
    return life, key, e


def main():
    import torch
    from hector_slam_amd import capi
    dev = torch.device("cuda", 0)
    B = 4096
    bp, bs, truth, init, init_pyr, pts, offs = bench.make_inputs(0, B)
    m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, 1, device=0)
    m.setUpdateFactorFree(0.4)
    m.setUpdateFactorOccupied(0.9)
    m.build_map(bp, bs)
    if "--shuffle" in sys.argv:  # the same scans in a random order: does the slow region move with the data?
        order = np.random.default_rng(3).permutation(B)
        scans = [pts[offs[i]:offs[i + 1]] for i in order]
        pts = np.concatenate(scans)
        offs = np.concatenate([[0], np.cumsum([len(s_) for s_ in scans])]).astype(offs.dtype)
        init = init[order]
    d_init, d_pts, d_offs = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (init, pts, offs))
    pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    cov = torch.zeros((B, 9), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream()
    for rep in range(4):
        m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS, pose.data_ptr(), cov.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
    if "--sustained" in sys.argv:  # 300 launches back to back first: the clock the kernel gets in a steady bench loop
        for rep in range(300):
            m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS, pose.data_ptr(), cov.data_ptr(), s.cuda_stream)
    prev = None
    for rep in range(2):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS, pose.data_ptr(), cov.data_ptr(), s.cuda_stream)
        b_.record()
        torch.cuda.synchronize()
        print(f"== launch {rep}: event time {a_.elapsed_time(b_) * 1e3:.1f} us")
        prev = analyse(cov.cpu().numpy().view(np.uint32), B, prev)


if __name__ == "__main__":
    main()
