"""The device-side pose gather (hsm_exchange_*, hector_slam_amd/csrc/pose_exchange.h) WITHOUT a GPU: its protocol -- layout,
epoch tags, flow control by mailbox depth -- runs between real processes over shared memory (tests/cpp/exchange_model.cpp
includes the very header the HIP kernels use), started by the ranks of a gloo process group that carries the shared-memory
names the way the GPU path carries its IPC handles.  The GPU tests (tests/test_gpu_exchange.py) run the kernels themselves."""
import json
import os
import socket
import subprocess
import sys
import uuid

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def model_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("xmodel") / "exchange_model"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "hector_slam_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "exchange_model.cpp"), "-o", str(out), "-lrt"], check=True)
    return str(out)


def _rank(rank, world, port, binary, total_rows, cols, depth, lag, epochs, seed, force, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    name = f"/hsm_xmodel_{uuid.uuid4().hex[:12]}_{rank}"
    try:
        names = [None] * world
        dist.all_gather_object(names, name)  # the GPU path: all_gather_object of the 64-byte IPC handles
        r = subprocess.run([binary, str(rank), str(world), str(total_rows), str(cols), str(depth), str(lag), str(epochs), str(seed),
                            "1" if force else "0"] + names, capture_output=True, text=True, timeout=240)
        rec = json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip() else {"error": r.stderr[-500:]}
        rec["rc"] = r.returncode
        dist.barrier()  # nobody unlinks a mailbox a peer may still read
        q.put(rec)
    finally:
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass
        dist.destroy_process_group()


def _run(binary, world, total_rows, cols, depth, lag, epochs, seed, force=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_rank, args=(r, world, port, binary, total_rows, cols, depth, lag, epochs, seed, force, q)) for r in range(world)]
    [p.start() for p in ps]
    recs = [q.get(timeout=300) for _ in range(world)]
    [p.join(60) for p in ps]
    return sorted(recs, key=lambda r: r.get("rank", -1))


@pytest.mark.parametrize("world,lag,depth,total_rows", [(2, 1, 4, 64), (2, 0, 2, 64), (3, 1, 4, 50), (2, 2, 6, 33)])
def test_protocol_between_processes(model_binary, world, lag, depth, total_rows):
    """ranks that drift (rank 0 fast, the last one slow, random stalls) never find a later epoch's rows in a buffer they
    have not unpacked, lose no value and time out nowhere -- equal and unequal shards, lag 0 / 1 / 2 at the minimum depth"""
    recs = _run(model_binary, world, total_rows, 3, depth, lag, 600, seed=11)
    assert len(recs) == world
    for r in recs:
        assert r["rc"] == 0 and r["violations"] == 0 and r["timeouts"] == 0 and r["wrong_values"] == 0 and r["refused_posts"] == 0, r
        assert r["posted"] == r["waited"] == 600, r


def test_a_mailbox_one_buffer_too_shallow_is_caught(model_binary):
    """negative control: lag 1 on a mailbox of depth 2 (the runtime refuses it; the model is forced): the fast rank's posts
    overwrite rows the slow rank has not unpacked, and the check sees it"""
    recs = _run(model_binary, 2, 64, 3, 2, 1, 400, seed=5, force=True)
    assert sum(r["violations"] for r in recs) > 0, recs


def test_the_runtime_rule_refuses_that_depth():
    """pose_exchange.h's rule, as the host runtime applies it: depth >= 2 + 2 lag"""
    src = r'''
#include "pose_exchange.h"
#include <stdio.h>
int main() {
  using namespace hsm;
  int bad = 0;
  for (int lag = 0; lag < 4; ++lag) {
    const int d = exchange_min_depth(lag);
    bad += d != 2 + 2 * lag;
    // a rank that keeps to `lag`: before the launch that posts e it has waited for e - 1 - lag
    for (unsigned long long e = 1; e < 50; ++e) {
      const unsigned long long waited = e > (unsigned long long)lag + 1 ? e - 1 - lag : 0;
      bad += !exchange_post_is_safe(e, waited, d);          // allowed at the minimum depth
      if (e > (unsigned long long)lag + 2) bad += exchange_post_is_safe(e, waited - 1, d);  // one more epoch ahead: refused
    }
  }
  ExchangeLayout L{2, 10, 3, 4};
  bad += L.buffer_of(5) != 1 * 30 || L.buffer_of(8) != 0 || L.granules() != 120;
  bad += !exchange_carries(exchange_pack(0xdeadbeefu, 0x100000007ull), 7) || exchange_value(exchange_pack(0xdeadbeefu, 9)) != 0xdeadbeefu;
  bad += exchange_carries(0, 1);  // a zero-filled mailbox matches no epoch >= 1 ...
  bad += !exchange_carries(exchange_pack(1u, 0x100000000ull), 0x100000000ull);  // ... and epoch 2^32 (tag 0) finds tag 2^32 - depth there, not 0
  printf("%d\n", bad);
  return bad != 0;
}
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.run(["g++", "-std=c++17", "-I", os.path.join(ROOT, "hector_slam_amd", "csrc"), os.path.join(d, "t.cpp"), "-o", os.path.join(d, "t")], check=True)
        r = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
