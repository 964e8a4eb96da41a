#!/usr/bin/env python3
"""Per-kernel duration summary from a rocprofv3 --kernel-trace run, with MIN and MEDIAN beside the mean.

rocprofv3's own `*_kernel_stats.csv` gives the mean over EVERY call of the process — bench.py's untimed pre-roll and warm-up
sweeps included (cold clocks: up to 15 % longer), so `launches x mean` exceeds the timed step. This reads the
per-dispatch trace instead and reports, per kernel: calls, min, median, mean, max over all calls AND over the calls of the
TIMED region, in microseconds. `--sweeps PRE,TIMED,POST`: sweeps the process ran before (pre-roll + warm-up), inside and after
(single synchronised sweeps, detail pass) the timed region; a kernel's calls are split pro rata (calls / total sweeps each).

usage: kernel_trace_stats.py OUT.csv TRACE_DIR [--sweeps PRE,TIMED,POST]"""
import csv
import glob
import statistics
import sys


def main():
    out, root = sys.argv[1], sys.argv[2]
    pre = timed = post = 0
    if "--sweeps" in sys.argv:
        pre, timed, post = (int(x) for x in sys.argv[sys.argv.index("--sweeps") + 1].split(","))
    per = {}
    for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "min_us", "median_us", "mean_us", "max_us", "timed_calls", "timed_min_us", "timed_median_us", "timed_mean_us"])
        for k, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
            v.sort()
            d = [x[1] / 1e3 for x in v]
            t = d
            total = pre + timed + post
            if total and len(d) % total == 0:      # (a kernel launched a whole number of times per sweep)
                per_sweep = len(d) // total
                t = d[pre * per_sweep:(pre + timed) * per_sweep]
            w.writerow([k, len(d), round(min(d), 2), round(statistics.median(d), 2), round(statistics.fmean(d), 2), round(max(d), 2),
                        len(t), round(min(t), 2), round(statistics.median(t), 2), round(statistics.fmean(t), 2)])


if __name__ == "__main__":
    main()
