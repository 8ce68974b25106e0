#!/bin/bash
# developer tool (round 6): bare stores and list kernel of BASELINE config 2 by the distance between its 62 output streams
# (PLLHIP_CLV_PAD_SITES: more slack behind every CLV; 8 sites = 1 KB).   bash tools/clv_pad_sweep.sh
export PLL_AMD_AUTO_MIRROR_MB=0 PLLHIP_DEVELOPER=1
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2; do
for pad in 0 32 512 1984 8128 16320 48512 24; do
  PLLHIP_CLV_PAD_SITES=$pad python3 bench.py --sites 1000000 --cpu-sites 0 --no-vary --no-c4 --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=r.get('box_ceiling') or {}
print('pad %6d sites (stride %11d B)  value %8.1f  launch %9.1f us  frac %.3f  bare stores %7.1f GB/s  of them %s' % ($pad, (1000064 + $pad) * 128, d['value'], r['avg_launch_us'], r['frac'], c.get('GBs', 0), r.get('frac_of_box_ceiling')))"
done; done
