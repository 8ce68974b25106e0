#!/bin/bash
export PLL_AMD_AUTO_MIRROR_MB=0   # (the device path is what is measured: no host mirrors kept for partitions below 64 MB)
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# On the GPU box: short op lists (partial traversals) at 1 M sites x 64 taxa -- per level, whole list with
# one launch
sites=${1:-1000000}
echo "== per level";                     PLLHIP_FUSED=0 python3 tools/partial_traversal_timing.py $sites
echo "== whole list";                    PLLHIP_FUSED=2 python3 tools/partial_traversal_timing.py $sites
