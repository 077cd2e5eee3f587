import sys, numpy as np, torch
sys.path.insert(0, '.')
from hector_slam_amd import capi, synth
for n_beams, size in ((16384, 2048), (1081, 1024)):
    import os
    os.environ["HSM_EXACT_DENSE_MIN"] = "512"
    sc = synth.make_scene(n_beams=n_beams, map_size=size, levels=3, resolution=0.05, n_build=20, n_query=2, room=(40.0, 30.0), seed=31)
    g = capi.MapRepMultiMap(sc.resolution, size, size, 3)
    g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    probe = torch.zeros(8, dtype=torch.int64, device="cuda")
    g.set_clock_probe(probe.data_ptr())
    g.debug_spec_stats(True)
    for k in range(3):
        g.matchData(sc.query_init[0], sc.query_scans[0])
    st = g.debug_spec_stats(False)
    torch.cuda.synchronize()
    t = probe.cpu().numpy()
    print(n_beams, g.last_launch_config()["kernel"], "phases (cycles) A,B,C,D,wait:", [int(t[i+1]-t[i]) for i in range(5)], "stats/match", [x/3 for x in st])
