#!/bin/bash
T='tests/test_golden.py::test_hip_matches_golden[fused-dna_balanced16_tipclv_site]'
echo "== default"; PLLHIP_FUSED_DEBUG=2 python -m pytest "$T" -x -q 2>&1 | grep -v "^E  " | tail -40
echo "== drain"; PLL_AMD_LIB=$PWD/build_drain/libpll_amd.so python -m pytest "$T" -x -q 2>&1 | tail -3
echo "== reload off"; PLLHIP_FUSED_RELOAD=0 python -m pytest "$T" -x -q 2>&1 | tail -3
