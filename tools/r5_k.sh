#!/bin/bash
mkdir -p gpurun_out/r5k
{
echo "tools/small_partitions_ab.sh 4 / 20 (SITES=\"2000 6000 12000\"), one box, round 5 (segments in the whole-list kernels, the estimate walks the longest segment):"
echo "4 states:"; SITES="2000 6000 12000" bash tools/small_partitions_ab.sh 4
echo "20 states:"; SITES="2000 6000 12000" bash tools/small_partitions_ab.sh 20
echo "tools/small_partitions_shapes.sh:"; bash tools/small_partitions_shapes.sh
} > gpurun_out/r5k/small_partitions.txt 2>&1; cat gpurun_out/r5k/small_partitions.txt
