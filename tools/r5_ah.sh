#!/bin/bash
export PLLHIP_DEVELOPER=1
mkdir -p gpurun_out/r5ah
{
for rep in 1 2 3; do
for sites in 50000 100000 400000; do
for g in 0 512 640; do
  if [ $g = 0 ]; then echo -n "default (3 per CU)       "; tools/newton_floor.bin 20 $sites | cut -c1-120
  else echo -n "PLLHIP_AA_GRID_CAP=$g   "; PLLHIP_AA_GRID_CAP=$g tools/newton_floor.bin 20 $sites | cut -c1-120; fi
done; done; done
} > gpurun_out/r5ah/aa_deriv_grid_sizes.txt 2>&1; cat gpurun_out/r5ah/aa_deriv_grid_sizes.txt
