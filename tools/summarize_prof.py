#!/usr/bin/env python
"""Condense a tools/gpu_profile.sh output directory into one text summary (kernel stats + PMC per kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("stats/**/*kernel_stats.csv"):
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i > 12:
                break
            print(",".join(c[:70] for c in row))
print()
print("== PMC counters: mean per dispatch, by kernel ==")
for f in find("calib_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[row.get("Kernel_Name", "?").split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("# calibration (4096^2 level = 16777216 cells; expected KB: rebuild_prob R 65536 W 65536, rebuild_quad R 65536 W 262144, pack_cells R 131072 W 131072, fill_level W 458752)")
    for k, cs in acc.items():
        if k.startswith("hsm::"):
            print("  ", k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0][:60]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("#", os.path.relpath(f, out))
    for k, cs in acc.items():
        if "gn_match" not in k and "update_" not in k and "likelihood" not in k:
            continue
        print("  ", k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
