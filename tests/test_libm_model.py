"""csrc/libm_exact.h (PRODUCT header: the device's sincosf / expf) against the host libm, on the CPU.

The header is plain C++ behind a macro, so the very source the kernels compile is built here with g++ and swept
against glibc's sincosf / expf.  The default run covers every 61st float bit pattern (70 M arguments, all
exponents, both signs, NaN/inf/denormals); `HSM_LIBM_SWEEP_STRIDE=1` sweeps all 2^32 (18 s on 8 cores; the result of
that run is committed as profiles/r02/libm_model_exhaustive_cpu.txt: 0 mismatches for both functions).

The model follows glibc's FMA ifunc variants (what any x86-64 CPU with FMA+AVX2 runs).  A host without FMA is not a valid
checker host -- glibc then evaluates the same polynomials unfused and the last bit may differ: the test FAILS there (and
oracle/pyoracle.py refuses to load), it does not skip.
"""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_has_fma():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                f = line.split()
                return "fma" in f and "avx2" in f
    except OSError:
        pass
    return False


@pytest.mark.parametrize("where", ["cpu", pytest.param("gpubox", marks=pytest.mark.gpu)])
def test_libm_model_equals_host_libm(tmp_path, where):
    assert _cpu_has_fma(), "host CPU without FMA/AVX2: glibc uses its unfused sincosf / expf variants; parity would be unpinned"
    exe = tmp_path / "libm_model_check"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-pthread", os.path.join(HERE, "cpp", "libm_model_check.cpp"),
                    "-o", str(exe), "-lm"], check=True)
    stride = os.environ.get("HSM_LIBM_SWEEP_STRIDE", "61")
    r = subprocess.run([str(exe), stride], capture_output=True, text=True)
    out = json.loads(r.stdout)
    assert out["sincosf_mismatches"] == 0 and out["expf_mismatches"] == 0, out
    assert out["checked"] >= (1 << 32) // int(stride)
    assert r.returncode == 0
