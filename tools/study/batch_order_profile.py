import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hector_slam_amd import capi, synth
B, N = 4096, 1081
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
sc = synth.make_scene(n_beams=N, resolution=0.05, n_build=60, n_query=B, seed=77, pad_to_full=True, map_size=2048, levels=1, room=(40.0, 30.0))
g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, device=0)
g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
g.build_map(sc.build_poses, sc.build_scans)
pts, offs = synth.pack_scans(sc.query_scans)
d_pts, d_offs, d_init = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(sc.query_init).to(dev)
d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
g.set_batch_order(capi.ORDER_MORTON)
for _ in range(300):
    g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, d_pose.data_ptr(), 0, s)
torch.cuda.synchronize()
