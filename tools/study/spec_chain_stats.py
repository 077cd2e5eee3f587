#!/usr/bin/env python
"""Acceptance statistics of the speculative-carry chains (hector_slam_amd/csrc/spec_chain.h) on REAL chains: the nine
product sequences of getCompleteHessianDerivs (OccGridMapUtil.h:76-98) of synthetic scans on maps built by the CPU oracle, at
the start pose and at the converged pose, dense (16 384 beams, configs[4]) and Hokuyo (1081 beams, configs[1]).  CPU only.
Builds tests/cpp/spec_chain_model.cpp, feeds it the chains, prints its JSON lines (one per input and segment count K).
usage: tools/study/spec_chain_stats.py [out.jsonl]"""
import json, math, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "study"))
import numpy as np
from hector_slam_amd import synth
from oracle import pyoracle
from binade_stats import products


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    pyoracle.build()
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "model")
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "hector_slam_amd", "csrc"),
                        os.path.join(ROOT, "tests", "cpp", "spec_chain_model.cpp"), "-o", exe], check=True)
        cases = [(16384, 2048, (40.0, 30.0), 30.0, 31, [16, 32, 64, 112]), (16384, 2048, (40.0, 30.0), 30.0, 77, [112]),
                 (1081, 1024, (40.0, 30.0), 30.0, 5, [8, 16, 32, 64]), (1081, 1024, (20.0, 15.0), 30.0, 6, [16, 32]), (360, 512, (20.0, 15.0), 30.0, 7, [8, 16])]
        for n_beams, size, room, rmax, seed, Ks in cases:
            sc = synth.make_scene(n_beams=n_beams, map_size=size, levels=1, resolution=0.05, n_build=30, n_query=4, room=room, seed=seed)
            o = pyoracle.Oracle("ho", sc.resolution, size, size, 1)
            o.set_update_factor_free(0.4); o.set_update_factor_occupied(0.9)
            o.build_map(sc.build_poses, sc.build_scans)
            for q in range(4):
                pts = sc.query_scans[q]
                for label, pose_w in (("start", sc.query_init[q]), ("converged", o.match(sc.query_init[q], pts)[0])):
                    pr = products(o, 0, o.map_coords_pose(0, pose_w), pts)
                    f = os.path.join(d, "p.bin")
                    np.ascontiguousarray(pr.T).tofile(f)
                    r = subprocess.run([exe, "file", f, str(len(pts))] + [str(k) for k in Ks], capture_output=True, text=True)
                    for ln in r.stdout.splitlines():
                        rec = json.loads(ln)
                        rec["input"] = f"{n_beams} beams, {size}^2 map, seed {seed}, scan {q}, {label} pose"
                        print(json.dumps(rec))
                        if out:
                            out.write(json.dumps(rec) + "\n")
                    if r.returncode != 0:
                        print("MODEL FAILED", r.stderr[-500:])
                        sys.exit(1)


if __name__ == "__main__":
    main()
