#!/bin/bash
# round 5, first GPU call: the new sharded + repeats + per-rate cases, and where a NEW 20-state list's time goes
mkdir -p gpurun_out/r5a
python3 -m pytest tests/test_gpu_sharded.py -x -q > gpurun_out/r5a/sharded.txt 2>&1; tail -3 gpurun_out/r5a/sharded.txt
python3 tools/aa_new_list_host_time.py > gpurun_out/r5a/new_list_host.txt 2>&1; tail -40 gpurun_out/r5a/new_list_host.txt
python3 tools/varying_lists_timing.py 20 200000 > gpurun_out/r5a/varying_20.txt 2>&1; cat gpurun_out/r5a/varying_20.txt
python3 tools/varying_lists_timing.py 4 1000000 > gpurun_out/r5a/varying_4.txt 2>&1; cat gpurun_out/r5a/varying_4.txt
