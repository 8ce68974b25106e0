#!/bin/bash
mkdir -p gpurun_out/r5b; export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/gpurun_out/r5b/trace20 -- python3 $root/tools/new_list_trace.py run 20 200000 > $root/gpurun_out/r5b/run20.txt 2>&1
python3 $root/tools/new_list_trace.py parse $root/gpurun_out/r5b/trace20 > $root/gpurun_out/r5b/parse20.txt 2>&1
cat $root/gpurun_out/r5b/parse20.txt; grep "call [01]" $root/gpurun_out/r5b/run20.txt | head -40
rm -rf $root/gpurun_out/r5b/trace20
cd $root
python3 bench.py --states 20 --sites 200000 --cpu-sites 0 --no-c4 > gpurun_out/r5b/bench_c3.json 2> gpurun_out/r5b/bench_c3.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r5b/bench_c3.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'], json.dumps(d['varying_lists'])[:1500])"
