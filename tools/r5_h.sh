#!/bin/bash
mkdir -p gpurun_out/r5h
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_repeats.py tests/test_gpu_thresholds.py -x -q -k "not full_size" > gpurun_out/r5h/tests.txt 2>&1; tail -3 gpurun_out/r5h/tests.txt
{
for st in 4 20; do for sites in 2000 12000; do tools/step_floor.bin $st $sites 3; done; done
echo "== PLLHIP_FUSE_REDUCE=0 (workgroup sums added by the host also for small grids)"
for st in 4 20; do for sites in 2000 12000; do PLLHIP_FUSE_REDUCE=0 tools/step_floor.bin $st $sites 3; done; done
} > gpurun_out/r5h/step_floor.txt 2>&1; cat gpurun_out/r5h/step_floor.txt
export TMPDIR=/tmp; root=$(pwd); cd /tmp
for st in 4 20; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sf$st -- $root/tools/step_floor.bin $st 2000 3 > /dev/null 2>&1
echo "== kernels of step_floor $st 2000 3 (under the profiler)"; python3 $root/tools/kernel_stats.py /tmp/sf$st 8
done > $root/gpurun_out/r5h/step_floor_kernels.txt 2>&1; cat $root/gpurun_out/r5h/step_floor_kernels.txt
