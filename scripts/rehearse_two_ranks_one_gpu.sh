#!/bin/bash
# bench.py's N = 2 control flow on a ONE-GPU box: two ranks, both on device 0, the gradient all-reduce staged through the host over
# gloo (dp.HostComm: RCCL refuses two ranks on one GPU). Exercises on real hardware what the gloo tests exercise on the CPU double:
# two library contexts, hooks, row panels, the eager-vs-segments trial, max-over-ranks timing, the single-GPU figures, the
# strong-scaling secondary and the line's shape. The NUMBERS mean nothing (host-staged collective, two processes sharing one GPU).
# usage: rehearse_two_ranks_one_gpu.sh OUTDIR [bench flags]
set -e
out=${1:-gpurun_out/rehearse2}; shift || true
mkdir -p $out
port=29541
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port MDHIP_DEVICE=0 MDHIP_BENCH_HOST_COMM=1 HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout -k 10 500 python3 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --detail $out/detail.json "$@" > $out/rank$r.out 2> $out/rank$r.err &
  pids[$r]=$!
done
rc=0
for r in 0 1; do wait ${pids[$r]} || rc=$?; done
echo "rc=$rc"
tail -c 1800 $out/rank0.out
exit $rc
