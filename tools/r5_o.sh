#!/bin/bash
mkdir -p gpurun_out/r5o
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s step %8.1f us, lnL call %.1f us wall, lnl kernel %s' % ('$1', d['ms_per_step']*1e3, d['api_calls']['edge_loglikelihood_ms_wall']['median']*1e3, d['kernels'].get('lnl')))"; }
{
for rep in 1 2 3; do
for lib in "" build/ab_pipe/libpll_amd.so; do
  PLL_AMD_LIB=$lib python3 bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-c4 --no-vary 2>/dev/null | line "C2 ${lib:-this build (rolled, one sub-step ahead)}"
done; done
for lib in "" build/ab_pipe/libpll_amd.so; do
  PLL_AMD_LIB=$lib python3 bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-c4 --no-vary --sites 500000 --taxa 200 --tree random 2>/dev/null | line "C5 shape ${lib:-this build}"
  PLL_AMD_LIB=$lib python3 bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-c4 --no-vary --sites 100000 2>/dev/null | line "100 k sites ${lib:-this build}"
  PLL_AMD_LIB=$lib python3 bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-c4 --no-vary --total-sites 8000000 --taxa 128 2>/dev/null | line "C4 whole ${lib:-this build}"
done
} > gpurun_out/r5o/lnl_unroll_ab.txt 2>&1; cat gpurun_out/r5o/lnl_unroll_ab.txt
