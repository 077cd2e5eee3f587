import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from hector_slam_amd import capi
bp, bs, truth, init_l0, init_pyr, pts, offs = bench.make_inputs(0, 4096)[:7]
m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, 1)
m.setUpdateFactorFree(0.4); m.setUpdateFactorOccupied(0.9)
m.build_map(bp, bs)
scan = pts[offs[0]:offs[1]]
states = np.stack([m.getMapCoordsPose(0, p) for p in init_l0]).astype(np.float32)
for f, name in ((m.likelihood_states, "likelihood"), (m.residual_states, "residual")):
    f(0, states, scan)
    t = []
    for _ in range(20):
        a = time.perf_counter(); r = f(0, states, scan); t.append(time.perf_counter() - a)
    print(name, "4096 states x", scan.shape[0], "beams: host call median %.1f us" % (np.median(t) * 1e6), r[:3])
t = []
for _ in range(20):
    a = time.perf_counter(); cm, cw, lh = m.covariance_for_poses(0, states[:512], scan); t.append(time.perf_counter() - a)
print("covariance 512 poses (3584 evaluations): host call median %.1f us" % (np.median(t) * 1e6))
