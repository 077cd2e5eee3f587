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
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0][:60]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("#", os.path.relpath(f, out))
    for k, cs in acc.items():
        if "gn_match" not in k and "update_" not in k:
            continue
        print("  ", k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
