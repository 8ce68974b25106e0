#!/bin/bash
# VGPRs / SGPRs / scratch / occupancy of every kernel in one .hip file (compiler remarks; no GPU needed)
#   bash tools/kernel_resources.sh libpll_amd/csrc/hip/partials_fused.hip [name filter]
#   (KR_FLAGS: extra compiler flags, e.g. KR_FLAGS='-mllvm -amdgpu-mfma-vgpr-form' for partials_aa_fused.hip)
f=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Iinclude -Ilibpll_amd/csrc/hip $KR_FLAGS \
  -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | \
  grep -E "Function Name|SGPRs:|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | \
  paste - - - - - - | grep -E "$filt" | while IFS=$'\t' read -r n s v sc o l; do
    echo "$(echo "$n" | sed 's/Function Name: //' | c++filt | cut -c1-90) | $s | $v | $sc | $o"
  done
