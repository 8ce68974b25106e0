#!/bin/bash
mkdir -p gpurun_out/r5p
{
timeout 600 python3 tools/soak_shards.py 0 400
PLLHIP_SHARD_THREADS=0 timeout 300 python3 tools/soak_shards.py 1000 150
# four processes side by side, each with its own groups and threads
for w in 1 2 3 4; do timeout 600 python3 tools/soak_shards.py $((2000 + 300 * w)) 250 > gpurun_out/r5p/w$w.txt 2>&1 & done; wait; tail -1 gpurun_out/r5p/w?.txt
} > gpurun_out/r5p/soak_shards.txt 2>&1; tail -12 gpurun_out/r5p/soak_shards.txt
timeout 400 python3 tools/crash_soak.py --seconds 150 --workers 12 --log gpurun_out/r5p/crash_soak.log > gpurun_out/r5p/crash_soak.txt 2>&1; tail -3 gpurun_out/r5p/crash_soak.txt
timeout 400 python3 tools/crash_soak.py --seconds 120 --workers 12 --env PLL_AMD_DEVICES=0,0,0 --log gpurun_out/r5p/crash_soak_sharded.log > gpurun_out/r5p/crash_soak_sharded.txt 2>&1; tail -3 gpurun_out/r5p/crash_soak_sharded.txt
