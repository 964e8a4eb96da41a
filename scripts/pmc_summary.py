#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass) into per-kernel means.

usage: pmc_summary.py OUT.csv DIR_FETCH DIR_WRITE [...more dirs]
FETCH_SIZE / WRITE_SIZE are KiB per dispatch. On gfx950 FETCH_SIZE reports half
the bytes of a wide coalesced read (MI355X_MICROARCH.md §HBM), so
hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "hbm_bytes_per_launch_corrected"])
        for k, c in sorted(agg.items()):
            fe = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [0])), 1)
            wr = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [0])), 1)
            n = max(len(c.get("FETCH_SIZE", [])), len(c.get("WRITE_SIZE", [])))
            w.writerow([k, n, round(fe, 1), round(wr, 1), int((2 * fe + wr) * 1024)])


if __name__ == "__main__":
    main()
