#!/bin/bash
# On the GPU box (round 6): does storing a value that the list copies back later with the default cache policy
# (PLLHIP_AA_KEEP=1; opt-in since the end of round 6) instead of non-temporally (0) change where the copy comes from?  Root 1's list of
# tools/replay_pmc.py (three such copies).
export TMPDIR=/tmp PLLHIP_DEVELOPER=1 REPLAY_ROOTS=1
root=$(pwd); out=$root/gpurun_out/r6_keep; mkdir -p "$out"; cd /tmp
for keep in 0 1; do
  export PLLHIP_AA_KEEP=$keep
  echo "==== PLLHIP_AA_KEEP=$keep"
  n=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$((n + 1))
    rocprofv3 --pmc $set --output-format csv -d "$out/pass$n" -- python3 "$root/tools/replay_pmc.py" run > /dev/null 2> "$out/pass$n.err"
    python3 "$root/tools/replay_pmc.py" parse "$out/pass$n" | grep -v dispatches
    rm -rf "$out/pass$n"
  done
done
