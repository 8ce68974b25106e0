#!/bin/bash
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# On the GPU box: the 4-state whole-list launch across the partition shapes of BASELINE configs 2, 4 (shard and
# whole), 5 and a 256-taxon list: launch time, rate on the launch's own bytes, fraction of the 8 TB/s peak.
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-40s launch %8.2f ms  %6.1f GB/s  frac %.3f  value %.1f lnL %.4f' % ('$1', r['avg_launch_us']/1e3, r['achieved'], r['frac'], d['value'], d['lnl']))"; }
for shape in "--sites 1000000 --taxa 64" "--sites 1000000 --taxa 128" "--total-sites 8000000 --taxa 128" "--sites 2000000 --taxa 256" "--sites 500000 --taxa 200 --tree random" "--sites 1000000 --taxa 64"; do
  python3 bench.py $shape --cpu-sites 0 --steps 10 --warmup 2 --no-vary --no-c4 2>/dev/null | line "$shape"
done
