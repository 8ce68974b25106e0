#!/bin/bash
mkdir -p gpurun_out/r5j
SITES="2000 6000 12000" bash tools/small_partitions_ab.sh 4 > gpurun_out/r5j/small_4.txt 2>&1; cat gpurun_out/r5j/small_4.txt
SITES="2000 6000 12000" bash tools/small_partitions_ab.sh 20 > gpurun_out/r5j/small_20.txt 2>&1; cat gpurun_out/r5j/small_20.txt
{
echo "== C3: this build against -DPLLHIP_AF_PIPE=2 (build/ab_pipe)"; bash tools/ab_two_libs.sh build/ab_pipe/libpll_amd.so --no-vary --no-c4 --states 20 --sites 200000
echo "== random 200 x 100k"; bash tools/ab_two_libs.sh build/ab_pipe/libpll_amd.so --no-vary --no-c4 --states 20 --sites 100000 --taxa 200 --tree random
} > gpurun_out/r5j/ab_pipe2.txt 2>&1; cat gpurun_out/r5j/ab_pipe2.txt
