#!/bin/bash
# rocprofv3 kernel stats of the reduction lab -> gpurun_out/$1/{lab.log,kernel_stats.csv}; $2 = variant groups ("all", "cols,sum", ..)
set -e
tag=${1:-lab}; sel=${2:-all}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- scripts/lab/reduce_lab.bin 12 $sel > $out/lab.log 2>&1
cp $(find $out/prof -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
rm -rf $out/prof
python3 - $out/kernel_stats.csv <<'PY' | tee $out/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    n = r["Name"]
    if any(k in n for k in ("k_cols", "k_sum", "k_maskprod", "k_reduce", "k_finish", "k_ew_fast")):
        print("%9.2f us avg %9.2f min  x%-4s %s" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"], n[:110]))
PY
cat $out/lab.log | tail -70
