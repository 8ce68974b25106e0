#!/bin/bash
mkdir -p gpurun_out/r5i
( time python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r5i/gpu_suite.txt 2>&1; tail -6 gpurun_out/r5i/gpu_suite.txt
{
echo "== C3: this build against the same sources with -DPLLHIP_AF_PIPE=1 (build/ab_pipe)"; bash tools/ab_two_libs.sh build/ab_pipe/libpll_amd.so --no-vary --no-c4 --states 20 --sites 200000
echo "== random 200 x 100k"; bash tools/ab_two_libs.sh build/ab_pipe/libpll_amd.so --no-vary --no-c4 --states 20 --sites 100000 --taxa 200 --tree random
echo "== random 64 x 200k"; bash tools/ab_two_libs.sh build/ab_pipe/libpll_amd.so --no-vary --no-c4 --states 20 --sites 200000 --taxa 64 --tree random
} > gpurun_out/r5i/ab_pipe.txt 2>&1; cat gpurun_out/r5i/ab_pipe.txt
for st in 4 20; do for sites in 2000 12000; do tools/step_floor.bin $st $sites 3; done; done > gpurun_out/r5i/step_floor.txt 2>&1; cat gpurun_out/r5i/step_floor.txt
