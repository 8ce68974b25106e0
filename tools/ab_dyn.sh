#!/bin/bash
for d in 2 4 7 10 14 30; do echo "== dynamic rounds $d"; bash tools/ab_fused.sh dyn$d PLLHIP_FUSED_DYNAMIC_ROUNDS=$d 2>&1 | head -3; done
