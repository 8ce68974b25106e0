#!/bin/bash
mkdir -p gpurun_out/r5u
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_repeats.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r5u/tests.txt; cat gpurun_out/r5u/tests.txt
python3 tools/repeats_reidentify.py > gpurun_out/r5u/reidentify.txt 2>&1; cat gpurun_out/r5u/reidentify.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5u/prof -o re -- python3 tools/repeats_reidentify.py > gpurun_out/r5u/prof.log 2>&1
python3 tools/kernel_stats.py gpurun_out/r5u/prof 25 > gpurun_out/r5u/stats.txt 2>&1
python3 tools/kernel_timeline.py gpurun_out/r5u/prof 70 > gpurun_out/r5u/timeline.txt 2>&1
rm -rf gpurun_out/r5u/prof
cat gpurun_out/r5u/stats.txt; tail -70 gpurun_out/r5u/timeline.txt
for a in "--states 4" "--states 20 --sites 200000"; do python3 bench.py --site-repeats $a --cpu-sites 0 --no-vary --no-c4 --steps 10 2>&1 | tail -3; done > gpurun_out/r5u/bench_repeats.txt 2>&1; cat gpurun_out/r5u/bench_repeats.txt
