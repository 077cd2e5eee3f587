#!/usr/bin/env python
"""Register / LDS / scratch use of every kernel in the gfx950 code object (compiles every translation unit to
assembly and reads the AMDGPU metadata).  Usage: tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hector_slam_amd import build  # noqa: E402


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    extra = sys.argv[2:]
    s = build.device_asm(extra)
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
