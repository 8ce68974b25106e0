#!/bin/bash
# On the GPU box (round 6): what do the operands copied back from HBM cost the 20-state list kernel?  Tool build
# (-DPLLHIP_AF_TIMING -DPLLHIP_AF_NOTICKS, build/afexp), PLLHIP_AF_EXP: 0 as it is, 32 the copies are requested but not
# waited for (wrong results: timing only).
export PLLHIP_DEVELOPER=1
lib=build/afexp/libpll_amd.so
for shape in "--states 20 --sites 100000 --taxa 200 --tree random" "--states 20 --sites 200000 --taxa 64 --tree random" "--states 20 --sites 200000"; do
  for rep in 1 2; do
    for m in 0 32; do
      PLLHIP_AF_EXP=$m PLL_AMD_LIB=$lib python3 bench.py --steps 10 --warmup 2 --cpu-sites 0 --no-c4 --no-vary $shape 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s mask %2d: update_partials %.3f ms' % ('$shape', $m, d['api_calls']['update_partials_ms_hip_events']['median']))"
    done
  done
done
