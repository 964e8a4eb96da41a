#!/bin/bash
# usage: prof_ab.sh TAG WORKLOAD "BENCH FLAGS" LIB...   (LIB = "main" or a name under scripts/ab/)
# rocprofv3 kernel stats of bench.py with each library build in turn (selected by MDHIP_LIB_VARIANT: the product file is never touched)
tag=$1; wl=$2; flags=$3; shift 3
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for lib in "$@"; do
  if [ $lib = main ]; then unset MDHIP_LIB_VARIANT; else export MDHIP_LIB_VARIANT=$lib; fi
  for rep in 1 2; do
    d=$out/${wl}_${lib}_$rep
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline $flags > $d.log 2>&1 || exit 1
    cp $(find $d -name '*kernel_stats.csv' | head -1) $d.csv
    rm -rf $d
    echo "== $wl $lib rep $rep: $(grep -o '"value": [0-9.]*' $d.log | head -1)"
    python3 - $d.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('>(')[0]
    print("   %-90s calls %4s avg %9.1f us" % (n[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
  done
done
