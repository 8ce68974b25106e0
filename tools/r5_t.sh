#!/bin/bash
mkdir -p gpurun_out/r5t
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s step %8.1f us value %9.1f' % ('$1', d['ms_per_step']*1e3, d['value']))"; }
for rep in 1 2 3; do
python3 bench.py --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "one partition, 1,000,000 sites"
python3 bench.py --gpus 8 --in-process --devices 0,0,0,0,0,0,0,0 --total-sites 1000000 --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "eight shards of 125,000 sites on ONE device"
done > gpurun_out/r5t/eight.txt 2>&1; cat gpurun_out/r5t/eight.txt
