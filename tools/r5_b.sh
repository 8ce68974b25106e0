#!/bin/bash
mkdir -p gpurun_out/r5b; export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trace20 -- python3 $root/tools/new_list_trace.py run 20 200000 > $root/gpurun_out/r5b/run20.txt 2>&1
find /tmp/trace20 | head -20
python3 $root/tools/new_list_trace.py parse /tmp/trace20 > $root/gpurun_out/r5b/parse20.txt 2>&1
cat $root/gpurun_out/r5b/parse20.txt
cd $root
bash tools/size_sweep2.sh > gpurun_out/r5b/size_sweep_before.txt 2>&1; cat gpurun_out/r5b/size_sweep_before.txt
