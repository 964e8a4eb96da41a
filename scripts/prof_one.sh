#!/bin/bash
# usage: prof_one.sh TAG NAME bench-flags...   -> kernel stats + PMC traffic of one workload/mode (see prof_round.sh)
set -e
tag=$1; name=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
d=$out/ks_$name
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 3 "$@" > $out/${RTAG:-r3}_bench_${name}_under_rocprof.log 2>&1
cp $(find $d -name '*kernel_stats.csv' | head -1) $out/${RTAG:-r3}_${name}_kernel_stats.csv; rm -rf $d
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_${name}_$c -- python3 bench.py --no-secondary --no-cpu-baseline --steps 3 --warmup 1 "$@" > $out/pmc_${name}_$c.log 2>&1
done
python3 scripts/pmc_summary.py $out/${RTAG:-r3}_pmc_$name.csv $out/pmc_${name}_FETCH_SIZE $out/pmc_${name}_WRITE_SIZE
rm -rf $out/pmc_${name}_FETCH_SIZE $out/pmc_${name}_WRITE_SIZE
echo "$name done"
