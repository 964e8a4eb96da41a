#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the committed per-workload PMC summaries (profiles/r4_pmc_*.csv, written by
scripts/pmc_summary.py out of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes). bench.py quotes it as
`traffic_committed`.   usage: pmc_traffic.py [PROFILES_DIR]"""
import csv
import json
import os
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
SOURCE = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --no-secondary --steps 3 --warmup 1 --workload W "
          "[--lazy]` (scripts/prof_round.sh, round 4); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 FETCH_SIZE counts "
          "half of wide reads: MI355X_MICROARCH.md HBM)")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "", 1) if name.startswith("void ") else name.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):          # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and i > 0:
            return name[:i]
    return name


kernels = {}
for wl, tag in (("cfg2", ""), ("cfg3", ""), ("cfg3_lazy", ""), ("cfg4", " [cfg4]"), ("cfg4_lazy", " [cfg4_lazy]")):
    f = os.path.join(root, f"r4_pmc_{wl}.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        key = short(r["kernel"]) + tag
        if key in kernels:
            continue
        kernels[key] = {"hbm_bytes_per_launch": int(r["hbm_bytes_per_launch_corrected"]), "dispatches": int(r["dispatches"]),
                        "file": f"profiles/r4_pmc_{wl}.csv"}
json.dump({"_source": SOURCE, "kernels": kernels}, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
print(len(kernels), "kernels")
