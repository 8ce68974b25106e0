#!/bin/bash
mkdir -p gpurun_out/r5g
python3 -m pytest tests/test_gpu_sharded.py tests/test_gpu_thresholds.py -x -q > gpurun_out/r5g/tests.txt 2>&1; tail -3 gpurun_out/r5g/tests.txt
bash tools/shards_ab.sh > gpurun_out/r5g/shards_ab.txt 2>&1; cat gpurun_out/r5g/shards_ab.txt
export TMPDIR=/tmp; root=$(pwd); cd /tmp
for st in 4 20; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sf$st -- $root/tools/step_floor.bin $st 2000 3 > $root/gpurun_out/r5g/step_floor_trace_$st.txt 2>&1
f=$(find /tmp/sf$st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4,6,7 "$f" | sed 's/^"\([^("]*\)[^"]*"/\1/' | head -12 >> $root/gpurun_out/r5g/step_floor_trace_$st.txt
cat $root/gpurun_out/r5g/step_floor_trace_$st.txt
done
