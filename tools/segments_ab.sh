#!/bin/bash
export PLL_AMD_AUTO_MIRROR_MB=0   # (the device path is what is measured: no host mirrors kept for partitions below 64 MB)
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# On the GPU box: whole-list launches with and without (tile, segment) work items at the sizes an 8-way shard of the
# BASELINE configs lands on.   bash tools/segments_ab.sh [states ...]
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=d['api_calls']
print('states %2d sites %7d %-26s step %8.1f us  update_partials %8.1f us (events)  frac %.3f' % ($1, $2, '$3', d['ms_per_step']*1e3, a['update_partials_ms_hip_events']['median']*1e3, r['frac']))"; }
if [ $# -eq 0 ]; then set -- "4 20000" "4 31250" "4 50000" "4 62500" "4 100000" "4 125000" "4 250000" "4 1000000"; fi
for cfg in "$@"; do
  set -- $cfg
  for rep in 1 2; do
    for v in 0 8; do
      PLLHIP_FUSED_SEGMENTS=$v python3 bench.py --states $1 --sites $2 --cpu-sites 0 --no-vary --no-c4 --steps 50 2>/dev/null | line $1 $2 "PLLHIP_FUSED_SEGMENTS=$v"
    done
  done
done
