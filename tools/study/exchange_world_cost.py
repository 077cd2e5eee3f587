#!/usr/bin/env python
"""What a matcher launch that carries the pose exchange costs as the world grows -- measured on ONE device: `world` mailboxes
connected inside the process (peer-access form), `world` matcher launches per step on one stream, each posting its 4096 rows to
every mailbox and unpacking all world x 4096 rows of the step before.  Same data path as N ranks (minus the xGMI hop).
usage: tools/study/exchange_world_cost.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hector_slam_amd import capi, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, N = 4096, 1081
sc = synth.make_scene(n_beams=N, map_size=2048, levels=1, resolution=0.05, n_build=60, n_query=B, room=(40.0, 30.0), seed=1234)
g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, 1, device=0)
g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
g.build_map(sc.build_poses, sc.build_scans)
pts, offs = synth.pack_scans(sc.query_scans)
dev = torch.device("cuda", 0)
d_pts, d_offs, d_init = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(sc.query_init).to(dev)
stream = torch.cuda.current_stream()
s = stream.cuda_stream
plain = torch.zeros((B, 3), dtype=torch.float32, device=dev)


def timed(fn, n):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(best)[1]


plain_launch = lambda: g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, plain.data_ptr(), 0, s)
timed(plain_launch, 1000)  # (the engine clock needs tens of milliseconds of load to settle)
base = timed(lambda: g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, plain.data_ptr(), 0, s), steps)
print("no exchange: %.2f us per launch" % base, g.last_launch_config()["kernel"], flush=True)
ref = plain.cpu().numpy().copy()
for world in (1, 2, 4, 8, 16):
    total = world * B
    xs = [capi.PoseExchange(r, world, total, 3, depth=4, device=0) for r in range(world)]
    for x in xs:
        x.connect_local(xs)
    local = [torch.zeros((B, 3), dtype=torch.float32, device=dev) for _ in range(world)]
    out = [torch.zeros((total, 3), dtype=torch.float32, device=dev) for _ in range(world)]
    state = {"e": 0}

    def step(fused=True):
        state["e"] += 1
        lands = state["e"] - 1 >= 1
        for r in range(world):
            if fused:
                g.match_batch_device_gather(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, local[r].data_ptr(), 0, xs[r], r * B, 1,
                                            out[r].data_ptr() if lands else 0, s)
            else:
                g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, local[r].data_ptr(), 0, s)
                xs[r].post_wait(local[r].data_ptr(), r * B, B, 1, out[r].data_ptr() if lands else 0, s)

    n = max(20, steps // world)
    base = timed(plain_launch, steps)  # (again, next to the legs it is compared with)
    t_f = timed(step, n) / world
    t_s = timed(lambda: step(False), n) / world
    torch.cuda.synchronize()
    for x in xs:
        x.check()
    ok = all(np.array_equal(out[r].cpu().numpy().view(np.uint32), np.tile(ref, (world, 1)).view(np.uint32)) for r in range(world))
    print("world %2d (%6d rows per mailbox): no exchange %.2f us per launch, carried by the matcher launch %.2f us per launch (+%.2f), stand-alone kernel behind it %.2f (+%.2f); gathered rows %s"
          % (world, total, base, t_f, t_f - base, t_s, t_s - base, "bit-identical" if ok else "DIFFER"), flush=True)
    for x in xs:
        x.close()
