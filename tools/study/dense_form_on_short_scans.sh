cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hector_slam_amd import capi, synth
for n_beams, size in ((1081, 2048), (720, 1024), (360, 1024)):
    sc = synth.make_scene(n_beams=n_beams, map_size=size, levels=3, resolution=0.05, n_build=20, n_query=2, room=(40.0, 30.0), seed=31)
    ref = None
    for env in (None, "200"):
        if env: os.environ["HSM_EXACT_DENSE_MIN"] = env
        else: os.environ.pop("HSM_EXACT_DENSE_MIN", None)
        g = capi.MapRepMultiMap(sc.resolution, size, size, 3)
        g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
        g.build_map(sc.build_poses, sc.build_scans)
        for k in range(50): p, c = g.matchData(sc.query_init[0], sc.query_scans[0])
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for k in range(200): p, c = g.matchData(sc.query_init[0], sc.query_scans[0])
            ts.append((time.perf_counter() - t0) / 200 * 1e6)
        bits = np.asarray(p, np.float32).view(np.uint32).tolist() + np.asarray(c, np.float32).view(np.uint32).ravel().tolist()
        ref = ref or bits
        cfg = g.last_launch_config()
        print(n_beams, len(sc.query_scans[0]), "HSM_EXACT_DENSE_MIN", env, cfg["kernel"], cfg["block"], "host call us: median %.1f" % sorted(ts)[2], "same bits" if bits == ref else "DIFFERENT", flush=True)
        g.close()
PY
