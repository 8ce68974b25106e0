#!/bin/bash
# A/B of one environment switch on ONE box: bash tools/ab_env.sh VAR "v1 v2 ..." [bench args...]
var=$1; vals=$2; shift 2
for rep in 1 2; do
for v in $vals; do
  env $var=$v python3 bench.py --cpu-sites 0 --steps 20 --warmup 3 --no-c4 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-26s %-34s launch_us %8.1f frac %.3f value %8.1f lnl %.6f' % ('$var=$v', '$*', r['avg_launch_us'], r['frac'], d['value'], d['lnl']))"
done; done
