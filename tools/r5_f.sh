#!/bin/bash
mkdir -p gpurun_out/r5f
{
for st in 4 20; do for sites in 2000 12000; do tools/step_floor.bin $st $sites 3; done; done
tools/step_floor.bin 4 2000 5; tools/step_floor.bin 20 2000 5
for st in 4 20; do PLLHIP_FUSED=0 tools/step_floor.bin $st 2000 3 | sed 's/^/FUSED=0 /'; PLLHIP_FUSED=2 tools/step_floor.bin $st 2000 3 | sed 's/^/FUSED=2 /'; done
} > gpurun_out/r5f/step_floor.txt 2>&1; cat gpurun_out/r5f/step_floor.txt
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s step %8.1f us value %9.1f per-rank %s' % ('$1', d['ms_per_step']*1e3, d['value'], d.get('per_rank_ms_per_step')))"; }
{
for rep in 1 2; do
python3 bench.py --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "one partition, 1,000,000 sites"
python3 bench.py --gpus 8 --in-process --devices 0,0,0,0,0,0,0,0 --total-sites 1000000 --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "eight shards of 125,000 sites on ONE device"
python3 bench.py --gpus 8 --in-process --devices 0,0,0,0,0,0,0,0 --total-sites 500000 --taxa 200 --tree random --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "C5 in eight shards of 62,500 sites on ONE device"
python3 bench.py --sites 500000 --taxa 200 --tree random --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "C5 one partition"
python3 bench.py --gpus 8 --in-process --devices 0,0,0,0,0,0,0,0 --total-sites 200000 --states 20 --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "C3 in eight shards of 25,000 sites on ONE device"
python3 bench.py --sites 200000 --states 20 --cpu-sites 0 --no-vary --no-c4 --steps 30 2>/dev/null | line "C3 one partition"
done
} > gpurun_out/r5f/eight_shards_one_device.txt 2>&1; cat gpurun_out/r5f/eight_shards_one_device.txt
