"""The restatement must reproduce the committed golden vectors (outputs of the reference's own
headers, tests/golden/make_golden.py) BIT-FOR-BIT.  Runs everywhere (no GPU, no /root/reference)."""
import os

import numpy as np
import pytest

from conftest import bits

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True, params=["cpu", pytest.param("gpubox", marks=pytest.mark.gpu)])
def where(request):
    """runs in the CPU suite and again (gpu-marked, no device needed) in the GPU box's `-m gpu` run"""
    return request.param


@pytest.fixture(scope="module")
def g1():
    return np.load(os.path.join(GOLD, "config1_181beam_256map.npz"))


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(GOLD, "pyramid_1081beam_512map.npz"))


def test_config1_single_level_181_beams(oracle_mod, g1):
    o = oracle_mod.Oracle("ho", float(g1["resolution"]), int(g1["map_size"]), int(g1["map_size"]), 1)
    o.upload_level(0, g1["logodds"], g1["update_index"])
    for q in range(4):
        pts, init = g1[f"q{q}_pts"], g1[f"q{q}_init"]
        pose, cov = o.match_level(0, init, pts, 5)
        assert np.array_equal(bits(pose), bits(g1[f"q{q}_pose"]))
        assert np.array_equal(bits(cov), bits(g1[f"q{q}_cov"]))
        for k in range(7):
            H, d = o.hessian_derivs(0, g1[f"q{q}_step_pose_map"][k], pts)
            assert np.array_equal(bits(H), bits(g1[f"q{q}_step_H"][k]))
            assert np.array_equal(bits(d), bits(g1[f"q{q}_step_dTr"][k]))
        # the reference converges: 5+1 GN steps pull a 15 cm / 0.05 rad error under 3 cm / 0.01 rad
        err = np.abs(pose - g1[f"q{q}_truth"])
        assert err[0] < 0.03 and err[1] < 0.03 and err[2] < 0.01


def test_pyramid_match_data(oracle_mod, g2):
    o = oracle_mod.Oracle("ho", float(g2["resolution"]), int(g2["map_size"]), int(g2["map_size"]), 3)
    for lvl in range(3):
        o.upload_level(lvl, g2[f"logodds{lvl}"], g2[f"update_index{lvl}"])
    for q in range(8):
        pose, cov = o.match(g2["init"][q], g2[f"q{q}_pts"])
        assert np.array_equal(bits(pose), bits(g2["pose"][q]))
        assert np.array_equal(bits(cov), bits(g2["cov"][q]))


def test_processor_trajectory(oracle_mod, g2):
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1081, map_size=512, levels=3, resolution=0.05, n_build=80, n_query=8,
                          room=(20.0, 15.0), seed=99)
    o = oracle_mod.Oracle("ho", sc.resolution, sc.map_size, sc.map_size, 3)
    o.set_update_factor_free(0.4)
    o.set_update_factor_occupied(0.9)
    o.proc_set_thresholds(0.0, 0.0)
    hint = sc.build_poses[0].copy()
    for t in range(12):
        o.proc_update(sc.build_scans[t], hint)
        hint, _ = o.proc_last_pose()
        assert np.array_equal(bits(hint), bits(g2["traj_pose"][t])), t
        hint = hint + (sc.build_poses[t + 1] - sc.build_poses[t])
    for lvl in range(3):
        lo, ui = o.download_level(lvl)
        assert np.array_equal(bits(lo), bits(g2[f"traj_logodds{lvl}"]))
        assert np.array_equal(ui, g2[f"traj_update_index{lvl}"])
