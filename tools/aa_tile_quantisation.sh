#!/bin/bash
# k_aa_fused: a workgroup tile is 32 sites and a CU holds two workgroups -- 16,384 sites per round of the chip.  How much
# of C3's distance from the write stream is the last, partly filled round?  bash tools/aa_tile_quantisation.sh
for sites in 196608 200000 212992 229376 393216 400000; do
  python3 bench.py --states 20 --sites $sites --cpu-sites 0 --no-vary --no-c4 --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('sites %7d  rounds %6.2f  update_partials %8.1f us  frac %.4f  value %.1f' % ($sites, $sites/16384.0, r['avg_launch_us'], r['frac'], d['value']))"
done
