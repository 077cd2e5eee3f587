"""Build the gfx950 shared library in-tree (hector_slam_amd/lib/libhector_mi355.so).

One translation unit (csrc/hector_mi355.hip + gn_match.h + map_update.h), compiled with
hipcc for gfx950 only.  -ffp-contract=off keeps every fp32 expression un-fused so the
per-beam arithmetic is bit-identical to the reference's x86-64 build (DESIGN.md "numerics").
hipcc cross-compiles without a GPU, so this also runs in the build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SRC = os.path.join(_PKG, "csrc", "hector_mi355.hip")
DEPS = [SRC, os.path.join(_PKG, "csrc", "gn_match.h"), os.path.join(_PKG, "csrc", "gn_match_exact.h"),
        os.path.join(_PKG, "csrc", "map_update.h"),
        os.path.join(_PKG, "csrc", "libm_exact.h"),
        os.path.join(_ROOT, "include", "hector_mi355", "capi.h")]
LIB = os.path.join(_PKG, "lib", "libhector_mi355.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function",
         # a kernel that misses the occupancy its __launch_bounds__ ask for fails the build (round 4 shipped four that did)
         "-Werror=pass-failed"]


def hipcc_path() -> str | None:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile if the library is missing or older than its sources; return its path."""
    if not force and not is_stale():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libhector_mi355.so (no CPU fallback exists)")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    tmp = f"{LIB}.{os.getpid()}.tmp"  # several ranks of one job may build at once: private temp, atomic rename
    cmd = [hipcc] + FLAGS + ["-I", os.path.join(_ROOT, "include"), "-I", os.path.join(_PKG, "csrc"),
                             SRC, "-o", tmp, "-ldl"]  # dlopen of librccl (part of libc since glibc 2.34; explicit for older ones)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
