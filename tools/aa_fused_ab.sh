#!/bin/bash
# On the GPU box: the 20-state whole-list kernel (default) against the per-level launches (PLLHIP_FUSED=0), ms per update
for cfg in "--taxa 64 --sites 200000" "--taxa 64 --sites 200000 --tree random" "--taxa 200 --sites 100000 --tree random" "--taxa 64 --sites 100000 --tip-clv" "--taxa 128 --sites 100000" "--taxa 64 --sites 50000" "--taxa 64 --sites 20000" "--taxa 64 --sites 200000 --no-scalers"; do
  for f in 1 0; do
    PLLHIP_FUSED=$f python3 bench.py --steps 10 --warmup 2 --cpu-sites 0 --no-c4 --no-vary --states 20 $cfg 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-48s %s: update_partials %.3f ms  lnL %.6f' % ('$cfg', 'whole list' if $f else 'per level ', d['api_calls']['update_partials_ms_hip_events']['median'], d['lnl']))"
  done
done
