#!/bin/bash
mkdir -p gpurun_out/r5d
python3 -m pytest tests/test_gpu_thresholds.py tests/test_gpu_aa_whole_list.py -x -q > gpurun_out/r5d/tests_a.txt 2>&1; tail -5 gpurun_out/r5d/tests_a.txt
python3 -m pytest tests/test_gpu_soak_at_size.py tests/test_gpu_baseline_configs.py -x -q > gpurun_out/r5d/tests_b.txt 2>&1; tail -5 gpurun_out/r5d/tests_b.txt
bash tools/segments_ab.sh "20 12500" "20 25000" "20 50000" "20 100000" "20 200000" > gpurun_out/r5d/segments_ab_20.txt 2>&1; cat gpurun_out/r5d/segments_ab_20.txt
python3 bench.py --cpu-sites 0 --no-c4 > gpurun_out/r5d/bench_c2.json 2>gpurun_out/r5d/bench_c2.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r5d/bench_c2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], json.dumps(d['varying_lists'])[:900])"
