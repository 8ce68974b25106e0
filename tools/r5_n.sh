#!/bin/bash
mkdir -p gpurun_out/r5n
python3 -m pytest tests/test_gpu_repeats.py tests/test_gpu_sharded.py -x -q > gpurun_out/r5n/tests.txt 2>&1; tail -3 gpurun_out/r5n/tests.txt
{
for rep in 1 2; do
echo "-- classes by hash table (default)"; python3 tools/repeats_reidentify.py 2>/dev/null
echo "-- classes by radix sort (PLLHIP_REPEATS_SORT=1)"; PLLHIP_REPEATS_SORT=1 python3 tools/repeats_reidentify.py 2>/dev/null
done
echo "-- first evaluation (every node identified) + steady state, C5 shape"
for v in 0 1; do PLLHIP_REPEATS_SORT=$v python3 bench.py --sites 500000 --taxa 200 --tree random --site-repeats --cpu-sites 0 --no-c4 --steps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PLLHIP_REPEATS_SORT=$v first evaluation %.1f ms, step %.3f ms' % (d['first_evaluation_ms'], d['ms_per_step']))"; done
} > gpurun_out/r5n/repeats_reidentify.txt 2>&1; cat gpurun_out/r5n/repeats_reidentify.txt
