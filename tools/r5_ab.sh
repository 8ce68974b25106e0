#!/bin/bash
export PLLHIP_DEVELOPER=1
mkdir -p gpurun_out/r5ab
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s lnl kernel %6.2f us, lnL call wall %6.1f us (median), step %7.1f us' % ('$1', d['kernels']['lnl']['avg_us'], d['api_calls']['edge_loglikelihood_ms_wall']['median']*1e3, d['ms_per_step']*1e3))"; }
{
for rep in 1 2 3; do
for g in 0 8192 768 2048; do
  if [ $g = 0 ]; then python3 bench.py --steps 20 --cpu-sites 0 --no-c4 --no-vary 2>/dev/null | line "C2 default (1024)"
  else PLLHIP_LNL_GRID=$g python3 bench.py --steps 20 --cpu-sites 0 --no-c4 --no-vary 2>/dev/null | line "C2 PLLHIP_LNL_GRID=$g"; fi
done; done
} > gpurun_out/r5ab/lnl_grid2.txt 2>&1; cat gpurun_out/r5ab/lnl_grid2.txt
