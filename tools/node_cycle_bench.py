#!/usr/bin/env python
"""Latency of the drop-in where the ROS node sits: tests/cpp/slam_driver.cpp drives the reference's UNCHANGED
HectorSlamProcessor::update() (matchData + updateByScan + onMapUpdated, every step updates the map) -- once compiled
against the reference tree (CPU) and once against the facade (MI355X); every call is timed inside the C++ driver.
configs[1]-shaped: 1081-beam scans, 3-level 1024/512/256 pyramid.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hector_slam_amd import synth  # noqa: E402
from test_facade_dropin import GPU_BIN, REF_BIN, write_scenario  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    sc = synth.make_scene(n_beams=1081, map_size=1024, levels=3, resolution=0.05, n_build=steps + 1, n_query=1,
                          room=(40.0, 30.0), seed=11)
    out = {"workload": f"HectorSlamProcessor::update per scan, {steps} scans of 1081 beams, 3-level 1024^2 pyramid, zero update "
                       "thresholds (every step = matchData + updateByScan on 3 levels + onMapUpdated)"}
    with tempfile.TemporaryDirectory() as d:
        scen = os.path.join(d, "s.bin")
        write_scenario(scen, sc, steps, hooks=0, min_dist=0.0, min_ang=0.0, mwm_at=())
        for name, exe in (("mi355x_dropin", GPU_BIN), ("reference_cpu", REF_BIN)):
            tf = os.path.join(d, name + ".json")
            env = dict(os.environ, SLAM_DRIVER_TIMING=tf)
            r = subprocess.run([exe, scen, os.path.join(d, name + ".bin")], env=env, capture_output=True, text=True)
            out[name] = json.load(open(tf)) if r.returncode == 0 and os.path.exists(tf) else {"error": r.stderr[-300:]}
    if "median_us" in out.get("mi355x_dropin", {}) and "median_us" in out.get("reference_cpu", {}):
        out["speedup_median"] = out["reference_cpu"]["median_us"] / out["mi355x_dropin"]["median_us"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
