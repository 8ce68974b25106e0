#!/bin/bash
export PLLHIP_DEVELOPER=1
mkdir -p gpurun_out/r5ac
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s lnl kernel %7.2f us, lnL call wall %7.1f us (median), step %8.1f us' % ('$1', d['kernels']['lnl']['avg_us'], d['api_calls']['edge_loglikelihood_ms_wall']['median']*1e3, d['ms_per_step']*1e3))"; }
{
for rep in 1 2; do
for sites in 125000 250000 500000 2000000; do
for g in 0 768 1024 1536; do
  if [ $g = 0 ]; then python3 bench.py --sites $sites --steps 20 --cpu-sites 0 --no-c4 --no-vary 2>/dev/null | line "$sites sites, default grid"
  else PLLHIP_LNL_GRID=$g python3 bench.py --sites $sites --steps 20 --cpu-sites 0 --no-c4 --no-vary 2>/dev/null | line "$sites sites, PLLHIP_LNL_GRID=$g"; fi
done; done; done
for g in 0 1024; do
  if [ $g = 0 ]; then python3 bench.py --total-sites 8000000 --taxa 128 --steps 5 --cpu-sites 0 --no-c4 --no-vary 2>/dev/null | line "8000000 sites x 128 taxa, default grid"
  else PLLHIP_LNL_GRID=$g python3 bench.py --total-sites 8000000 --taxa 128 --steps 5 --cpu-sites 0 --no-c4 --no-vary 2>/dev/null | line "8000000 sites x 128 taxa, PLLHIP_LNL_GRID=$g"; fi
done
} > gpurun_out/r5ac/lnl_grid_sizes.txt 2>&1; cat gpurun_out/r5ac/lnl_grid_sizes.txt
