#!/bin/bash
# On the GPU box: the soaks on the round's final library ("re-run the at-size soak on the final library").
export PLLHIP_DEVELOPER=1 PLL_AMD_AUTO_MIRROR_MB=0
out=gpurun_out/${1:-r6_soaks}; mkdir -p $out
{
python3 tools/soak_aa_fused_at_size.py 40000 1500 100000 200 20
python3 -c "
import sys; sys.path.insert(0, 'tools'); import soak_aa_fused_at_size as s
sys.exit(s.run(first=42000, count=400, sites=100000, T=200, states=20, rate_scalers=True))"
PLLHIP_AA_TI_MFMA=0 python3 tools/soak_aa_fused_at_size.py 43000 500 100000 200 20
python3 tools/soak_aa_fused_at_size.py 44000 800 30000 64 20
python3 tools/soak_aa_fused_at_size.py 45000 1500 100000 200 4
python3 -c "
import sys; sys.path.insert(0, 'tools'); import soak_aa_fused_at_size as s
sys.exit(s.run(first=47000, count=400, sites=100000, T=200, states=4, rate_scalers=True))"
python3 tools/soak_aa_fused.py 20000 1500
PLLHIP_AA_TI_MFMA=0 python3 tools/soak_aa_fused.py 22000 1000
python3 tools/soak_fused.py 20000 1000
python3 tools/soak.py 20000 300
python3 tools/soak_repeats_at_size.py 40000 60
} > $out/soaks_final_library.log 2>&1
grep -E "^soak|scaling certificate|MISMATCH|Error|error" $out/soaks_final_library.log
