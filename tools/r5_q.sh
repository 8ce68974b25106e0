#!/bin/bash
mkdir -p gpurun_out/r5q
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_thresholds.py tests/test_gpu_sharded.py -x -q -k "not full_size" > gpurun_out/r5q/tests.txt 2>&1; tail -2 gpurun_out/r5q/tests.txt
bash tools/size_sweep2.sh > gpurun_out/r5q/size_sweep.txt 2>&1; cat gpurun_out/r5q/size_sweep.txt
{
echo "== C2: this build against round 4's"; bash tools/ab_two_libs.sh build/ab_head/libpll_amd.so --no-vary --no-c4
} > gpurun_out/r5q/ab_c2.txt 2>&1; cat gpurun_out/r5q/ab_c2.txt
for st in 4; do for sites in 2000 12000; do tools/step_floor.bin $st $sites 3; tools/step_floor.bin $st $sites 5; done; done > gpurun_out/r5q/step_floor.txt 2>&1; cat gpurun_out/r5q/step_floor.txt
export TMPDIR=/tmp; root=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python3 $root/bench.py --steps 20 --warmup 2 --cpu-sites 0 --no-vary --no-c4 --sites 62500 > /dev/null 2>&1
python3 $root/tools/kernel_stats.py /tmp/kp 4 > $root/gpurun_out/r5q/kernels_62500.txt 2>&1; cat $root/gpurun_out/r5q/kernels_62500.txt
