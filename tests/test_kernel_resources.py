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
def device_asm():
    from hector_slam_amd import build
    if build.hipcc_path() is None:
        pytest.skip("hipcc not found")
    return build.device_asm()  # every translation unit of the library, with the library's own flags


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
        m = re.search(r"gn_match_cached_kernelILi(\d+)ELi\d+ELi\d+ELi(\d+)ELb[01]E", k)
        waves = int(m.group(1)) * int(m.group(2))                # scans per workgroup x waves per scan
        per_cu = 20 if int(m.group(2)) == 2 else 16              # five / four waves per SIMD
        if int(m.group(2)) == 2:
            assert v["vgpr"] <= 96, (k, v)                        # the two-wave form is built for five waves per SIMD
        assert (per_cu // waves) * v["lds"] <= 160 * 1024, (k, v)  # the workgroups of one CU share its LDS


def test_counted_waits_are_static_in_the_peeled_step(device_asm):
    """the waits of the peeled first step must have folded to immediates (s_waitcnt vmcnt(n) in straight-line code): the
    quad-layout 17-beam kernel then contains the schedule's values (endpoints 4 ahead: 4..8 while they stream in, 2 / 1
    for a texel with the next endpoint + gather or only the next gather behind it)"""
    m = re.search(r"^_ZN3hsm22gn_match_cached_kernelILi4ELi17ELi1ELi1ELb0EEEvNS_11MatchParamsE:(.*?)^\.Lfunc_end", device_asm, re.S | re.M)
    assert m
    body = m.group(1)
    seen = {int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)}
    assert {0, 1, 2, 4, 5, 6, 7, 8} <= seen, sorted(seen)
    assert "scratch_" not in body


# ---- the inline-asm gather contract, checked on the ISA ------------------------------------------------------------------
# The texel-cache kernels issue their gathers from inline asm into registers the compiler believes are already written, and
# wait for them with counted s_waitcnt.  That is only correct if NOTHING touches a destination register between the load
# and the wait that covers it -- no copy (v_mov, v_accvgpr_write), no spill, no use as any operand.  The walker below keeps
# the queue of vector-memory operations in flight (loads return in issue order; stores count in vmcnt too on gfx9) and
# flags every instruction that names a register of a load still in flight.
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(text):
    out = set()
    for a, lo, hi in _REG.findall(text):
        if a:
            out.add(int(a))
        else:
            out.update(range(int(lo), int(hi) + 1))
    return out


def gather_contract_violations(body):
    """body: the kernel's assembly text.  Returns a list of (line number, instruction, registers) violations."""
    lines = [ln.split(";")[0].strip() for ln in body.split("\n")]
    lines = [ln for ln in lines if ln and not ln.startswith((".", "//"))]
    inflight = []  # [(set of destination VGPRs, text)], oldest first
    bad = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        op = ln.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ln)
            if m:
                n = int(m.group(1))
                # `vmcnt(1); s_branch 2f; 1: vmcnt(0); 2:` is ONE wait whose weaker arm may be the one taken
                nxt = lines[i + 1:i + 5]
                if n == 1 and len(nxt) >= 4 and nxt[0].startswith("s_branch") and nxt[2].startswith("s_waitcnt") and "vmcnt(0)" in nxt[2]:
                    i += 4
                del inflight[:max(0, len(inflight) - n)]
        elif op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            # the textual successor of an unconditional jump is not its control-flow successor: whatever is in flight here (a
            # load the COMPILER issued at the bottom of a loop and waits for at its top -- the inline-asm gathers of the
            # contract sit in straight-line code, apart from the two-armed wait handled above) is not in flight there
            inflight = []
        elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            dst = ln.split(None, 1)[1].split(",")[0]
            used = _vregs(ln.split(",", 1)[1]) if "," in ln else set()
            for regs, what in inflight:
                if regs & (used | _vregs(dst)):
                    bad.append((i, ln, sorted(regs & (used | _vregs(dst)))))
            inflight.append((_vregs(dst), ln))
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "flat_atomic")):
            used = _vregs(ln)
            for regs, what in inflight:
                if regs & used:
                    bad.append((i, ln, sorted(regs & used)))
            inflight.append((set(), ln))
        elif not op.endswith(":"):
            used = _vregs(ln)
            for regs, what in inflight:
                if regs & used:
                    bad.append((i, ln, sorted(regs & used)))
        i += 1
    return bad


def test_gather_contract_walker_catches_a_broken_sequence():
    ok = """
    global_load_dwordx4 v[10:13], v20, s[2:3]
    v_add_f32_e32 v1, v2, v3
    global_load_dwordx4 v[14:17], v21, s[2:3]
    s_waitcnt vmcnt(1)
    v_mul_f32_e32 v4, v10, v11
    s_waitcnt vmcnt(0)
    v_mul_f32_e32 v5, v14, v15
    """
    assert gather_contract_violations(ok) == []
    for broken in ("v_mov_b32_e32 v30, v12", "v_accvgpr_write_b32 a0, v10", "scratch_store_dwordx4 off, v[10:13], off",
                   "v_mul_f32_e32 v4, v14, v1"):
        text = ok.replace("s_waitcnt vmcnt(1)", broken + "\n    s_waitcnt vmcnt(1)")
        assert gather_contract_violations(text), broken
    # a consumer placed behind a wait that does not cover its load (the newest one is still in flight)
    text = ok.replace("v_mul_f32_e32 v4, v10, v11", "v_mul_f32_e32 v4, v14, v11")
    assert gather_contract_violations(text)


def texel_cache_kernels(asm):
    """every instantiation of the two kernels that issue their gathers from inline asm"""
    return sorted(set(re.findall(r"^(_ZN3hsm(?:22gn_match_cached_kernel|28gn_match_exact_cached_kernel)\w+):", asm, re.M)))


def test_no_register_of_an_inflight_gather_is_touched(device_asm):
    """EVERY instantiation in the default library (round-3 verdict: two were guarded): the quad and plane layouts, four and
    eight scans per workgroup, 9 and 17 rows, the relaxed arithmetic, and the three exact-order forms"""
    names = texel_cache_kernels(device_asm)
    assert len(names) >= 17, names
    for need in ("gn_match_cached_kernelILi4ELi17ELi1ELi1ELb0E", "gn_match_cached_kernelILi4ELi17ELi1ELi1ELb1E", "gn_match_cached_kernelILi8ELi17ELi1ELi1ELb0E",
                 "gn_match_cached_kernelILi4ELi9ELi1ELi1ELb0E", "gn_match_cached_kernelILi4ELi17ELi2ELi1ELb0E",
                 "gn_match_exact_cached_kernelILi4ELi17ELi13ELb0E", "gn_match_exact_cached_kernelILi4ELi17ELi15ELb0E", "gn_match_exact_cached_kernelILi4ELi9ELi9ELb0E", "gn_match_exact_cached_kernelILi4ELi5ELi5ELb0E",
                 "gn_match_exact_cached_kernelILi4ELi13ELi13ELb0E", "gn_match_exact_cached_kernelILi4ELi13ELi7ELb1E",
                 "gn_match_exact_cached_kernelILi4ELi13ELi13ELb1E", "gn_match_exact_cached_kernelILi4ELi17ELi15ELb1E",
                 "gn_match_exact_cached_kernelILi4ELi17ELi6ELb1E", "gn_match_exact_cached_kernelILi4ELi9ELi9ELb1E", "gn_match_exact_cached_kernelILi4ELi5ELi5ELb1E"):
        assert any(need in n for n in names), (need, names)
    for kernel in names:
        m = re.search(r"^" + re.escape(kernel) + r":(.*?)^\.Lfunc_end", device_asm, re.S | re.M)
        assert m, kernel
        body = m.group(1)
        assert len(re.findall(r"global_load_dwordx[24]", body)) >= 5, kernel  # the gathers are there (x4 texels, x2 plane pairs)
        bad = gather_contract_violations(body)
        assert not bad, (kernel, bad[:5])


# ---- achieved occupancy of EVERY matcher instantiation the launch helpers can reach -----------------------------------------
# Round 4 shipped four exact single-scan instantiations that reached three waves per SIMD where their __launch_bounds__ asked
# for four (a compiler warning nobody looked at).  Since round 5 the build fails on that warning (-Werror=pass-failed) and this
# test states the table: what each form is built for, against the `; Occupancy:` the compiler reports for it.
def occupancies(asm):
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"^(_ZN3hsm\w+):\s*;\s*@\1.*?; Occupancy: (\d+)", asm, re.S | re.M)}


def designed_waves_per_simd(name):
    m = re.search(r"15gn_match_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELb([01])E", name)
    if m:  # gn_match.h: four, except the exact-order teams of two / four wavefronts (a group of five rounds alive: three)
        wps, spb, exact = int(m.group(1)), int(m.group(2)), m.group(3) == "1"
        return 3 if (exact and spb == 1 and wps in (2, 4)) else 4
    m = re.search(r"22gn_match_cached_kernelILi\d+ELi\d+ELi\d+ELi(\d+)ELb[01]E", name)
    if m:
        return 5 if int(m.group(1)) > 1 else 4
    m = re.search(r"28gn_match_exact_cached_kernelILi\d+ELi\d+ELi(\d+)ELb([01])E", name)
    if m:  # chain-wavefront forms: five per SIMD (the compiler's figure includes the LDS: four workgroups of five wavefronts),
        # four for the full-texel-cache instantiations that are launched up to two workgroups per CU
        return (5 if 5 * int(m.group(1)) + 50 <= 96 else 4) if m.group(2) == "1" else 4
    if "20gn_match_coop_kernel" in name:
        return 1  # one workgroup per CU by design (K <= 64 workgroups on 256 CUs)
    return None


def test_chain_wavefront_forms_leave_room_for_six_wavefronts_per_simd(device_asm):
    """the chain-wavefront form (five wavefronts per workgroup) is placed three times per CU -- whatever SIMDs the first two
    workgroups' odd wavefronts landed on -- only if a SIMD holds SIX of its wavefronts: the dispatcher wants room for
    ceil(5 / 4) = 2 more on EVERY SIMD (tools/study/ubench_wg_placement.hip, profiles/r05/README.md 9).  512 / 6 -> 80 VGPRs."""
    ks = {k: v for k, v in kernels(device_asm).items() if re.search(r"28gn_match_exact_cached_kernelILi\d+ELi\d+ELi\d+ELb1E", k)}
    assert len(ks) == 6, sorted(ks)
    two_per_cu = [k for k in ks if "ILi4ELi17ELi15ELb1E" in k or "ILi4ELi13ELi13ELb1E" in k]  # launched up to two per CU: 2/1/1/1 leaves
    assert len(two_per_cu) == 2                                                                # two slots on every SIMD at four per SIMD
    for k, v in ks.items():
        assert v["vgpr"] <= (128 if k in two_per_cu else 80) and v["scratch"] == 0, (k, v)
        assert 3 * v["lds"] <= 160 * 1024, (k, v)


def test_build_fails_on_a_missed_occupancy_target():
    from hector_slam_amd import build
    assert "-Werror=pass-failed" in build.STRICT_FLAGS
    # checked builds (this suite, __graft_entry__.build(), the round-end script) keep it an error; only the lazily-run
    # user-side build may fall back to a build without the check (with a warning)
    assert os.environ.get("HSM_BUILD_STRICT") == "1"


def test_every_matcher_instantiation_reaches_its_designed_occupancy(device_asm):
    occ = occupancies(device_asm)
    assert len(occ) > 100
    seen = 0
    for name, waves in occ.items():
        want = designed_waves_per_simd(name)
        if want is None:
            continue
        seen += 1
        assert waves >= want, (name, waves, want)
    assert seen >= 80, seen  # the team forms (6 widths x 2 layouts x BPL), the texel-cache forms, the exact forms, the dense matcher
    # the forms the default mode launches: exact single scan on four wavefronts, the exact batch form, the exact dense team
    for need in ("15gn_match_kernelILi4ELi1ELi1ELi0ELb1E", "15gn_match_kernelILi4ELi1ELi2ELi0ELb1E", "15gn_match_kernelILi16ELi1ELi2ELi0ELb1E",
                 "28gn_match_exact_cached_kernelILi4ELi17ELi13ELb0E"):
        assert any(need in n for n in occ), need
