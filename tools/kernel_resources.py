#!/usr/bin/env python
"""Register / LDS / scratch use of every kernel in the gfx950 code object (compiles the translation unit with
-save-temps into a scratch directory and reads the AMDGPU metadata of the assembly).  Usage: tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hector_slam_amd import build  # noqa: E402


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    extra = sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        cmd = [build.hipcc_path()] + [f for f in build.FLAGS if f != "-shared"] + extra + [
            "-c", "-save-temps", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "hector_slam_amd", "csrc"),
            build.SRC, "-o", os.path.join(d, "x.o")]
        subprocess.run(cmd, check=True, cwd=d, stderr=subprocess.DEVNULL)
        s = open(os.path.join(d, "hector_mi355-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: +\d+", s, re.S):
        blk = m.group(0)
        name = re.search(r"\.name: +(\S+)", blk).group(1)
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("hsm::", "")
        if flt and flt not in dn:
            continue
        g = lambda k: re.search(r"\." + k + r": +(\d+)", blk).group(1)
        print(f"{dn[:100]:100s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} "
              f"scratch {g('private_segment_fixed_size'):>5s}")


if __name__ == "__main__":
    main()
