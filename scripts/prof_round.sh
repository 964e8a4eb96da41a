#!/bin/bash
# Round evidence for profiles/: rocprofv3 kernel stats of bench.py per workload and mode, and HBM traffic
# (separate --pmc passes for FETCH_SIZE and WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes).
# usage: prof_round.sh TAG   -> gpurun_out/TAG/{${RTAG:-r4}_*_kernel_stats.csv, ${RTAG:-r4}_*_kernel_minmedian.csv, ${RTAG:-r4}_pmc_*.csv, ${RTAG:-r4}_bench_*.json}
set -e
tag=${1:-r2prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, bench flags
  name=$1; shift
  d=$out/ks_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 3 "$@" > $out/${RTAG:-r4}_bench_${name}_under_rocprof.log 2>&1
  cp $(find $d -name '*kernel_stats.csv' | head -1) $out/${RTAG:-r4}_${name}_kernel_stats.csv
  # min / median beside the mean, and the timed region apart from the cold pre-roll + warm-up sweeps (bench.py: PREROLL + 3 warm-up,
  # 20 timed, then 10 single synchronised sweeps [+ 5 detail sweeps for cfg3 / cfg4])
  python3 scripts/kernel_trace_stats.py $out/${RTAG:-r4}_${name}_kernel_minmedian.csv $d --sweeps $SWEEPS
  rm -rf $d
  echo "kernel stats $name done"
}
pmc() {  # name, bench flags: two passes
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_${name}_$c -- python3 bench.py --no-secondary --no-cpu-baseline --steps 3 --warmup 1 "$@" > $out/pmc_${name}_$c.log 2>&1
  done
  python3 scripts/pmc_summary.py $out/${RTAG:-r4}_pmc_$name.csv $out/pmc_${name}_FETCH_SIZE $out/pmc_${name}_WRITE_SIZE
  rm -rf $out/pmc_${name}_FETCH_SIZE $out/pmc_${name}_WRITE_SIZE
  echo "pmc $name done"
}
SWEEPS=15,20,10 run cfg2 --workload cfg2
SWEEPS=23,20,15 run cfg3 --workload cfg3
SWEEPS=23,20,15 run cfg3_lazy --workload cfg3 --lazy
SWEEPS=13,20,15 run cfg4 --workload cfg4
SWEEPS=13,20,15 run cfg4_lazy --workload cfg4 --lazy
SWEEPS=43,20,10 run cfg5 --workload cfg5
pmc cfg4 --workload cfg4
pmc cfg4_lazy --workload cfg4 --lazy
pmc cfg3_lazy --workload cfg3 --lazy
pmc cfg3 --workload cfg3
pmc cfg2 --workload cfg2
python3 bench.py > $out/${RTAG:-r4}_bench_default.json 2> $out/${RTAG:-r4}_bench_default.err
echo all done
