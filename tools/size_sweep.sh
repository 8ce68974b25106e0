#!/bin/bash
export PLL_AMD_AUTO_MIRROR_MB=0   # (the device path is what is measured: no host mirrors kept for partitions below 64 MB)
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# BASELINE config 2's shape (and config 3's) at smaller site counts: where the fixed cost of a call shows.
#   bash tools/size_sweep.sh
for st in 4 20; do
for sites in 5000 10000 20000 50000 100000 200000 500000; do
  python3 bench.py --states $st --sites $sites --cpu-sites 0 --no-vary --no-c4 --steps 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=d['api_calls']
print('states %2d sites %7d  step %8.1f us  update_partials %8.1f us (events)  lnl call %6.1f us  frac %.3f  value %.1f  %s' % ($st, $sites, d['ms_per_step']*1e3, a['update_partials_ms_hip_events']['median']*1e3, a['edge_loglikelihood_ms_wall']['median']*1e3, r['frac'], d['value'], r['kernel'][:50]))"
done; done
