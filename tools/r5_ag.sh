#!/bin/bash
export PLLHIP_DEVELOPER=1
mkdir -p gpurun_out/r5ag
{
for rep in 1 2 3; do
for g in 0 256 512 1024 1536; do
  if [ $g = 0 ]; then echo -n "default                  "; tools/newton_floor.bin 20 200000 | cut -c1-250
  else echo -n "PLLHIP_AA_GRID_CAP=$g  "; PLLHIP_AA_GRID_CAP=$g tools/newton_floor.bin 20 200000 | cut -c1-250; fi
done; done
} > gpurun_out/r5ag/aa_grid.txt 2>&1; cat gpurun_out/r5ag/aa_grid.txt
