"""The speculative-carry form of the reference's fp32 chains (hector_slam_amd/csrc/spec_chain.h -- the header the device kernel
gn_match_spec_kernel uses) against the literal sequential loop, on the CPU: tests/cpp/spec_chain_model.cpp computes every chain both
ways and demands the same bits -- adversarial random chains (cancellation, ties, zeros, denormals, powers of two, sawtooth through
binades) and real ones (the nine product sequences of getCompleteHessianDerivs on an oracle-built map).  The GPU tests compare the
kernel itself with the reference (tests/test_gpu_exact_parity.py)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = tmp_path_factory.mktemp("specmodel") / "spec_chain_model"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "hector_slam_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "spec_chain_model.cpp"), "-o", str(exe)], check=True)
    return str(exe)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_shift_rule_is_exact_on_adversarial_chains(model, seed):
    r = subprocess.run([model, "random", str(seed), "2500"], capture_output=True, text=True)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and rec["mismatches"] == 0, (rec, r.stderr[-500:])
    # the cases do exercise every path: trivial candidates, real shifts, every kind of re-run
    assert rec["shifted"] > 10000 and rec["candidate_was_the_carry"] > 10000 and rec["rerun_out_of_range"] > 1000 and rec["rerun_not_a_multiple_or_tie"] > 1000, rec


def test_real_chains_accept_most_boundaries(model, oracle_mod, pyramid_scene, tmp_path):
    """the nine chains of real scans: bit-identical, and most boundaries are accepted (that is the speed; tools/study/spec_chain_stats.py
    has the full statistics, including the two-candidate 'collapse' variant that almost never applies)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools", "study"))
    from binade_stats import products
    from conftest import make_oracle
    sc = pyramid_scene
    o = make_oracle(oracle_mod, "ho", sc)
    tot = acc = 0
    for q in range(4):
        pts = sc.query_scans[q]
        pr = products(o, 0, o.map_coords_pose(0, sc.query_init[q]), pts)
        f = tmp_path / f"p{q}.bin"
        np.ascontiguousarray(pr.T).tofile(f)
        r = subprocess.run([model, "file", str(f), str(len(pts)), "16", "33"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-500:]
        for ln in r.stdout.splitlines():
            rec = json.loads(ln)
            assert rec["mismatches"] == 0
            tot += rec["boundaries"]
            acc += rec["candidate_was_the_carry"] + rec["shifted"]
            assert rec["two_candidate_collapse"] <= 0.2 * rec["two_candidate_tried"]
    assert acc / tot > 0.6, (acc, tot)
