#!/bin/bash
# Evidence for the native storage-only-dtype kernels (csrc/narrow.hip): kernel durations (rocprofv3 --kernel-trace --stats) and HBM
# traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes) of `int8 * int8`,
# `float16 + float16` and `sum(int8)` on 2**30 elements.   usage: prof_narrow.sh TAG -> gpurun_out/TAG/r4_{narrow_kernel_stats,pmc_narrow}.csv
set -e
tag=${1:-narrowprof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for c in "int8 * int8" "float16 + float16" "sum(int8) -> int64"; do
  n=$(echo "$c" | tr -c 'a-z0-9' '_')
  export NARROW_ONLY="$c"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$n -- python3 scripts/narrow_bench.py > $out/narrow_$n.log 2>&1
  python3 scripts/kernel_trace_stats.py $out/r4_narrow_${n}_minmedian.csv $out/ks_$n
  rm -rf $out/ks_$n
  for k in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $k --output-format csv -d $out/pmc_${n}_$k -- python3 scripts/narrow_bench.py > $out/pmc_${n}_$k.log 2>&1
  done
  python3 scripts/pmc_summary.py $out/r4_pmc_narrow_$n.csv $out/pmc_${n}_FETCH_SIZE $out/pmc_${n}_WRITE_SIZE
  rm -rf $out/pmc_${n}_FETCH_SIZE $out/pmc_${n}_WRITE_SIZE
  echo "$c done"
done
