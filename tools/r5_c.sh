#!/bin/bash
mkdir -p gpurun_out/r5c
python3 -m pytest tests/test_gpu_thresholds.py -x -q -k "segments and not 20-" > gpurun_out/r5c/seg_tests.txt 2>&1; tail -5 gpurun_out/r5c/seg_tests.txt
python3 -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -x -q -k "not full_size" > gpurun_out/r5c/api_tests.txt 2>&1; tail -3 gpurun_out/r5c/api_tests.txt
bash tools/segments_ab.sh > gpurun_out/r5c/segments_ab_4.txt 2>&1; cat gpurun_out/r5c/segments_ab_4.txt
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trace20 -- python3 $root/tools/new_list_trace.py run 20 200000 > $root/gpurun_out/r5c/run20.txt 2>&1
python3 $root/tools/new_list_trace.py parse /tmp/trace20 > $root/gpurun_out/r5c/parse20.txt 2>&1
cat $root/gpurun_out/r5c/parse20.txt
