#!/bin/bash
# the whole-list kernel across partition shapes: roofline fraction by taxa x sites (one box, one process each)
#   bash tools/shape_sweep.sh > profiles/<tag>_shape_sweep.txt
echo "# k_dna_fused, 4 states x 4 rates, balanced trees, per-site scalers: fraction of 8 TB/s on the kernel's own bytes"
echo "# (CLVs + scale buffers in GB in brackets; beyond 8 GB the counts are stored non-temporally)"
printf "%-8s" "taxa"; for s in 100000 250000 500000 1000000 2000000; do printf "%-22s" "$s sites"; done; echo
for t in 16 32 64 128 256; do
  printf "%-8s" $t
  for s in 100000 250000 500000 1000000 2000000; do
    python3 bench.py --cpu-sites 0 --no-c4 --steps 10 --warmup 2 --taxa $t --sites $s 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    gb=($t-2)*$s*132/1e9
    print('%-22s' % ('%.3f %6.1f us [%4.1f]' % (r['frac'], r['avg_launch_us'], gb)), end='')
except Exception as e:
    print('%-22s' % 'failed', end='')"
  done; echo
done
