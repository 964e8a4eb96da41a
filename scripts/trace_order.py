#!/usr/bin/env python3
"""Per-dispatch kernel durations in launch order from a rocprofv3 --kernel-trace csv (last N rows).  usage: trace_order.py DIR [N [SKIP_LAST]]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # rows to drop from the end first
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in (rows[-n - skip:-skip] if skip else rows[-n:]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  {name}")
    prev_end = e
