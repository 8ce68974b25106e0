#!/bin/bash
# Five bench.py processes in a row on one box (BASELINE config 2): how much of the spread between runs is the process,
# how much the box.  bash tools/c2_process_spread.sh
for i in 1 2 3 4 5; do
python3 bench.py --cpu-sites 0 --no-vary --no-c4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i', d['value'], d['ms_per_step'], d['roofline']['frac'], d['ramp'], d['api_calls']['update_partials_ms_hip_events'])"
done
