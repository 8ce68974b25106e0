#!/bin/bash
# On the GPU box: 4 states x 8 rate categories -- the whole-list kernel (PLLHIP_FUSED=2 / default) against per-level launches
for f in 0 2; do
  PLLHIP_FUSED=$f python3 bench.py --rate-cats 8 --sites 500000 --cpu-sites 0 --steps 10 --warmup 2 --no-vary --no-c4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('PLLHIP_FUSED=$f rate_cats 8, 500 k sites x 64 taxa: step %.3f ms, %.1f M site-updates/s, roofline frac %.3f (%s), lnL %.6f' % (d['ms_per_step'], d['value'], r['frac'], r['kernel'][:40], d['lnl']))"
done
