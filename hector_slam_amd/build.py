"""Build the gfx950 shared library in-tree (hector_slam_amd/lib/libhector_mi355.so).

One object per translation unit under csrc/ (compiled in parallel with hipcc for gfx950 only, re-compiled only when one of
its own sources changed), linked into one shared library.  -ffp-contract=off keeps every fp32 expression un-fused so the
per-beam arithmetic is bit-identical to the reference's x86-64 build (DESIGN.md "numerics").
hipcc cross-compiles without a GPU, so this also runs in the build container.

A kernel that misses the occupancy its __launch_bounds__ ask for is a build ERROR where the build is checked (HSM_BUILD_STRICT=1:
__graft_entry__.build(), the test suite, the round-end script); the lazily-run user-side build (capi.load_library on a machine
without a prebuilt library) retries without -Werror=pass-failed and warns, so that another compiler version costs performance,
not the import.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_CSRC = os.path.join(_PKG, "csrc")
_CAPI = os.path.join(_ROOT, "include", "hector_mi355", "capi.h")


def _c(*names):
    return [os.path.join(_CSRC, n) for n in names]


# translation unit -> the files it includes (its own staleness test)
_COMMON = _c("gn_match.h", "libm_exact.h", "hsm_host.h", "hsm_ctx.h") + [_CAPI]
UNITS = {
    "hector_mi355.hip": _c("hector_mi355.hip", "gn_match_spec.h", "spec_chain.h", "map_update.h") + _COMMON,
    "match_exact_cached.hip": _c("match_exact_cached.hip", "gn_match_exact.h") + _COMMON,
    "match_teams.hip": _c("match_teams.hip") + _COMMON,
    "pose_exchange.hip": _c("pose_exchange.hip", "pose_exchange.h", "hsm_host.h") + [_CAPI],
}
SRC = os.path.join(_CSRC, "hector_mi355.hip")
DEPS = sorted({d for deps in UNITS.values() for d in deps})
LIB = os.path.join(_PKG, "lib", "libhector_mi355.so")
OBJ_DIR = os.path.join(_PKG, "lib", "obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
         "-Wall", "-Wno-unused-function"]
# a kernel that misses the occupancy its __launch_bounds__ ask for fails the build (round 4 shipped four that did)
STRICT_FLAGS = ["-Werror=pass-failed"]


def hipcc_path() -> str | None:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _obj(unit: str) -> str:
    return os.path.join(OBJ_DIR, os.path.splitext(unit)[0] + ".o")


def _unit_stale(unit: str) -> bool:
    o = _obj(unit)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in UNITS[unit])


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def _compile(hipcc: str, unit: str, strict: bool, verbose: bool) -> str:
    out = _obj(unit)
    tmp = f"{out}.{os.getpid()}.tmp"  # several ranks of one job may build at once: private temp, atomic rename
    base = [hipcc] + FLAGS + ["-I", os.path.join(_ROOT, "include"), "-I", _CSRC, "-c", os.path.join(_CSRC, unit), "-o", tmp]
    cmd = base + STRICT_FLAGS
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and not strict and "pass-failed" in (r.stderr or ""):
        print(f"hector_slam_amd.build: {unit}: a kernel missed an optimisation / occupancy target with this compiler "
              f"(-Werror=pass-failed); building without the check -- expect lower performance\n{r.stderr[-2000:]}", file=sys.stderr)
        r = subprocess.run(base, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout or "")
        sys.stderr.write(r.stderr or "")
        raise subprocess.CalledProcessError(r.returncode, cmd)
    if verbose and r.stderr:
        sys.stderr.write(r.stderr)
    os.replace(tmp, out)
    return out


def build_native(force: bool = False, verbose: bool = False, strict: bool | None = None) -> str:
    """Compile what is missing or older than its sources, link; return the library's path."""
    if not force and not is_stale():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libhector_mi355.so (no CPU fallback exists)")
    if strict is None:
        strict = os.environ.get("HSM_BUILD_STRICT", "0") == "1"
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [u for u in UNITS if force or _unit_stale(u)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(lambda u: _compile(hipcc, u, strict, verbose), todo))
    tmp = f"{LIB}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(u) for u in UNITS] + ["-o", tmp, "-ldl"]  # dlopen of librccl (part of libc since glibc 2.34; explicit for older ones)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    return LIB


def build_variant(dst: str, extra_flags: list[str], verbose: bool = False) -> str:
    """an alternative library (extra -D / -mllvm flags on every unit) for A/B runs: HSM_LIB=<dst>; objects in a private directory"""
    hipcc = hipcc_path()
    objdir = dst + ".obj"
    os.makedirs(objdir, exist_ok=True)

    def one(unit):
        o = os.path.join(objdir, os.path.splitext(unit)[0] + ".o")
        cmd = [hipcc] + FLAGS + STRICT_FLAGS + list(extra_flags) + ["-I", os.path.join(_ROOT, "include"), "-I", _CSRC, "-c", os.path.join(_CSRC, unit), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return o
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(one, UNITS))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", dst, "-ldl"], check=True)
    return dst


def device_asm(extra_flags: list[str] | None = None) -> str:
    """the gfx950 assembly of every unit, concatenated (tests/test_kernel_resources.py, tools/kernel_resources.py)"""
    import tempfile
    hipcc = hipcc_path()
    with tempfile.TemporaryDirectory() as d:
        def one(unit):
            out = os.path.join(d, unit + ".s")
            cmd = [hipcc] + [f for f in FLAGS if f != "-fPIC"] + STRICT_FLAGS + list(extra_flags or []) + [
                "-S", "--cuda-device-only", "-I", os.path.join(_ROOT, "include"), "-I", _CSRC, os.path.join(_CSRC, unit), "-o", out]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            return open(out).read()
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
            return "\n".join(ex.map(one, UNITS))


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True, strict="--no-strict" not in sys.argv))
