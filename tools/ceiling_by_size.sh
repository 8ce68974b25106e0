#!/bin/bash
# developer tool (round 6): the bare stores of a list (roofline.box_ceiling) and the list kernel by partition size, one box:
# slow boxes store config 2's list at 5.7 TB/s and config 4 whole (8 M sites) at 7.2 -- where does it change?
export PLL_AMD_AUTO_MIRROR_MB=0 PLLHIP_DEVELOPER=1
cd "$(dirname "$0")/.." || exit 1
for cfg in "--sites 250000" "--sites 500000" "--sites 1000000" "--sites 2000000" "--sites 4000000" "--sites 8000000" "--sites 1000000 --taxa 128" "--sites 1000000 --taxa 32" "--sites 1000000 --no-scalers"; do
  python3 bench.py $cfg --cpu-sites 0 --no-vary --no-c4 --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=r.get('box_ceiling') or {}
print('%-34s value %8.1f  launch %9.1f us  frac %.3f  bare stores %7.1f GB/s (%s passes of %.3f ms)  of them %s' % ('$cfg', d['value'], r['avg_launch_us'], r['frac'], c.get('GBs', 0), c.get('passes'), c.get('ms_per_pass', 0), r.get('frac_of_box_ceiling')))"
done
