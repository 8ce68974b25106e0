#!/bin/bash
mkdir -p gpurun_out/r5z
{
for rep in 1 2; do for st in 4 20; do for sites in 2000 12000; do tools/step_floor.bin $st $sites 3; done; done; done
tools/newton_floor.bin 4 500000
} > gpurun_out/r5z/step_floor.txt 2>&1; cat gpurun_out/r5z/step_floor.txt
