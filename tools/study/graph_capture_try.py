import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hector_slam_amd import capi, synth
B, N = 4096, 1081
dev = torch.device("cuda", 0)
sc = synth.make_scene(n_beams=N, resolution=0.05, n_build=60, n_query=B, seed=77, pad_to_full=True, map_size=2048, levels=1, room=(40.0, 30.0))
g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, device=0)
g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
g.build_map(sc.build_poses, sc.build_scans)
g.synchronize()
pts, offs = synth.pack_scans(sc.query_scans)
d_pts, d_offs, d_init = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(sc.query_init).to(dev)
d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
s = torch.cuda.Stream()
def launch():
    g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, d_pose.data_ptr(), 0, s.cuda_stream)
with torch.cuda.stream(s):
    for _ in range(300): launch()
torch.cuda.synchronize()
ref = d_pose.cpu().numpy().copy()
K = 20
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph, stream=s):
        for _ in range(K): launch()
    print("captured", K, "launches")
except Exception as e:
    print("capture failed:", repr(e)[:300]); sys.exit(0)
d_pose.zero_()
graph.replay(); torch.cuda.synchronize()
print("replay poses bit-identical:", np.array_equal(d_pose.cpu().numpy().view(np.uint32), ref.view(np.uint32)))
for name, fn in (("stream launches", lambda: [launch() for _ in range(K)]), ("graph replay", graph.replay)):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / (20 * K) * 1e6)
    print(name, "us per launch: %.2f (min %.2f)" % (sorted(ts)[2], min(ts)))
