#!/usr/bin/env python
"""Build alternative libhector_mi355 variants (extra -D / -mllvm flags) into hector_slam_amd/lib/variants/
for A/B runs: HSM_LIB=<path> python bench.py ...   usage: build_variants.py name:"flags" ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hector_slam_amd import build as b
out = os.path.join(os.path.dirname(b.LIB), "variants")
os.makedirs(out, exist_ok=True)
for spec in sys.argv[1:]:
    name, flags = spec.split(":", 1)
    print(b.build_variant(os.path.join(out, f"libhector_mi355_{name}.so"), flags.split()))
