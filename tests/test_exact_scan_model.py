"""The block-scan form of the reference's sequential fp32 sums (tools/study/exact_scan.h -- a study, not part of the shipped
library: measured on MI355X it does not beat the literal chain, profiles/r05/README.md) must at least be RIGHT: its host model
runs the header's element arithmetic through the device code's data flow (64 lanes x E elements, integer prefix scans, tie
parity, first-violation restart) and has to reproduce the literal loop bit for bit on adversarial data."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_block_scan_model_reproduces_the_literal_chain(tmp_path):
    exe = tmp_path / "exact_scan_model"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "tools", "study"),
                    os.path.join(ROOT, "tools", "study", "exact_scan_model.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe), "1200"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["mismatches"] == 0 and rec["cases"] == 2400
    assert rec["scan_iterations"] > 100000 and rec["ties"] > 100000 and rec["single_adds"] > 1000  # every path was taken
