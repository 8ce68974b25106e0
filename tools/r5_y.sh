#!/bin/bash
mkdir -p gpurun_out/r5y
{
for rep in 1 2; do
tools/newton_floor.bin 4 500000
tools/newton_floor.bin 4 1000000
tools/newton_floor.bin 20 200000
done
echo "== the same calls through ctypes (bench.py --newton 20)"
python3 bench.py --sites 500000 --taxa 200 --tree random --cpu-sites 0 --no-vary --no-c4 --steps 20 --newton 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5 shape:', d['newton'], d['api_calls'])"
} > gpurun_out/r5y/newton_floor.txt 2>&1; cat gpurun_out/r5y/newton_floor.txt
