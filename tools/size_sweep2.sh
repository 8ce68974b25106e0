#!/bin/bash
export PLL_AMD_AUTO_MIRROR_MB=0   # (the device path is what is measured: no host mirrors kept for partitions below 64 MB)
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# The sizes an 8-way shard of the BASELINE configs lands on (C2 125 k, C5 62.5 k, C3 25 k sites) and their neighbours:
# whole-list launches with fewer tiles than a few rounds of workgroup slots.
#   bash tools/size_sweep2.sh
for cfg in "4 20000" "4 31250" "4 50000" "4 62500" "4 100000" "4 125000" "4 250000" "20 12500" "20 25000" "20 50000" "20 100000"; do
  set -- $cfg
  python3 bench.py --states $1 --sites $2 --cpu-sites 0 --no-vary --no-c4 --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=d['api_calls']
print('states %2d sites %7d  step %8.1f us  update_partials %8.1f us (events)  lnl call %6.1f us  frac %.3f  value %.1f' % ($1, $2, d['ms_per_step']*1e3, a['update_partials_ms_hip_events']['median']*1e3, a['edge_loglikelihood_ms_wall']['median']*1e3, r['frac'], d['value']))"
done
