#!/bin/bash
export PLLHIP_DEVELOPER=1   # developer switches are honoured only under this one
# the lookup ops' tables made by k_af_prepare (PLLHIP_AA_LOOKUP_DIRECT=1, default) against six launches of the
# tabulating kernels ahead of every list (=0, round 3): interleaved, one box.  bash tools/aa_lookup_direct_ab.sh
exec bash tools/ab_env.sh PLLHIP_AA_LOOKUP_DIRECT "0 1" "--states 20 --sites 200000" "--states 20 --sites 100000 --taxa 200 --tree random" "--states 20 --sites 30000" "--states 20 --sites 30000 --taxa 200 --tree random"
