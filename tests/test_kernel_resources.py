"""Static guard on the gfx950 code object (no GPU needed: hipcc cross-compiles).  The texel-cache matcher issues its
loads from inline asm and waits for them with counted s_waitcnt: the compiler does not know that the destination
registers are written asynchronously, so a register-allocator spill or copy of one of them between the load and its wait
would silently use stale data.  The GPU parity tests would catch that; this test catches the precondition on the build
box: the throughput kernels must not spill at all and must fit the occupancy they are designed for."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    from hector_slam_amd import build
    hipcc = build.hipcc_path()
    if hipcc is None:
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa") / "device.s"
    cmd = [hipcc] + [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] + [
        "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "hector_slam_amd", "csrc"),
        build.SRC, "-o", str(out)]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out.read_text()


def kernels(asm):
    res = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: +\d+", asm, re.S):
        blk = m.group(0)
        name = re.search(r"\.name: +(\S+)", blk).group(1)
        g = lambda k: int(re.search(r"\." + k + r": +(\d+)", blk).group(1))  # noqa: E731
        res[name] = {"vgpr": g("vgpr_count"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")}
    return res


def test_no_kernel_spills(device_asm):
    ks = kernels(device_asm)
    assert len(ks) > 40
    spilled = {k: v for k, v in ks.items() if v["scratch"] != 0}
    assert not spilled, spilled


def test_texel_cache_matcher_fits_four_waves_per_simd(device_asm):
    ks = {k: v for k, v in kernels(device_asm).items() if "gn_match_cached_kernel" in k}
    assert ks
    for k, v in ks.items():
        assert v["vgpr"] <= 128 and v["scratch"] == 0, (k, v)   # 512 VGPRs per SIMD lane / 4 waves
        m = re.search(r"gn_match_cached_kernelILi(\d+)ELi\d+ELi\d+ELi(\d+)E", k)
        waves = int(m.group(1)) * int(m.group(2))                # scans per workgroup x waves per scan
        per_cu = 20 if int(m.group(2)) == 2 else 16              # five / four waves per SIMD
        if int(m.group(2)) == 2:
            assert v["vgpr"] <= 96, (k, v)                        # the two-wave form is built for five waves per SIMD
        assert (per_cu // waves) * v["lds"] <= 160 * 1024, (k, v)  # the workgroups of one CU share its LDS


def test_counted_waits_are_static_in_the_peeled_step(device_asm):
    """the waits of the peeled first step must have folded to immediates (s_waitcnt vmcnt(n) in straight-line code): the
    quad-layout 17-beam kernel then contains the schedule's values (endpoints 4 ahead: 4..8 while they stream in, 2 / 1
    for a texel with the next endpoint + gather or only the next gather behind it)"""
    m = re.search(r"^_ZN3hsm22gn_match_cached_kernelILi4ELi17ELi1ELi1EEEvNS_11MatchParamsE:(.*?)^\.Lfunc_end", device_asm, re.S | re.M)
    assert m
    body = m.group(1)
    seen = {int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)}
    assert {0, 1, 2, 4, 5, 6, 7, 8} <= seen, sorted(seen)
    assert "scratch_" not in body
