#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py cfg4 (eager, lazy) and cfg3 lazy -> gpurun_out/$1
set -e
tag=${1:-r2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for mode in eager lazy; do
  flag=""; [ $mode = lazy ] && flag="--lazy"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg4_$mode -- python3 bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $flag > $out/cfg4_$mode.log 2>&1
  cp $(find $out/cfg4_$mode -name '*kernel_stats.csv' | head -1) $out/cfg4_${mode}_kernel_stats.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg3_lazy -- python3 bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --lazy > $out/cfg3_lazy.log 2>&1
cp $(find $out/cfg3_lazy -name '*kernel_stats.csv' | head -1) $out/cfg3_lazy_kernel_stats.csv
find $out -name '*.db' -delete; find $out -name '*kernel_trace.csv' -delete
echo done
