#!/bin/bash
# On the GPU box: the counters of tools/replay_pmc.py, one pass per counter set (no trace domains next to --pmc).
export TMPDIR=/tmp PLLHIP_DEVELOPER=1
root=$(pwd); out=$root/gpurun_out/${1:-r6_replay}; mkdir -p "$out"; cd /tmp
n=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum"; do
  n=$((n + 1))
  rocprofv3 --pmc $set --output-format csv -d "$out/pass$n" -- python3 "$root/tools/replay_pmc.py" run > /dev/null 2> "$out/pass$n.err"
  python3 "$root/tools/replay_pmc.py" parse "$out/pass$n"
  rm -rf "$out/pass$n"
done
