#!/bin/bash
export PLLHIP_DEVELOPER=1
mkdir -p gpurun_out/r5ad
{
for rep in 1 2 3; do
for g in 0 256 384 768 1024; do
  for sites in 500000 1000000; do
  if [ $g = 0 ]; then echo -n "default (2 per CU)      "; tools/newton_floor.bin 4 $sites | cut -c1-110
  else echo -n "PLLHIP_DERIV_GRID=$g  "; PLLHIP_DERIV_GRID=$g tools/newton_floor.bin 4 $sites | cut -c1-110; fi
  done
done; done
} > gpurun_out/r5ad/deriv_grid.txt 2>&1; cat gpurun_out/r5ad/deriv_grid.txt
