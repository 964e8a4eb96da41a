#!/bin/bash
export MDHIP_EXPERIMENTS=1   # the library reads its experiment variables only behind this gate (csrc/md_options.h)
# A/B of the generated "evaluate + column-sum in one pass" kernel's geometry (MDHIP_SWEEP_NB bands per strip, MDHIP_SWEEP_RU rows
# per trip; 0 = the library's choice) on cfg4 --lazy: the pass's own time from kernel-attached timestamps.
for nb in 0 8 16 24 32 48 64; do for ru in 0 4 8; do
  r=$(MDHIP_SWEEP_NB=$nb MDHIP_SWEEP_RU=$ru python bench.py --workload cfg4 --lazy --no-secondary --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['kernels']['hbm_tail']['backward_pair']
print('%.1f passes/s   pass %.2f us   %.1f %% of 8 TB/s' % (d['value'], p['avg_launch_ms']*1e3, p['frac']*100))")
  echo "NB=$nb RU=$ru -> $r"
done; done
