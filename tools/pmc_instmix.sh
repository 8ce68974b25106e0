#!/bin/bash
# dynamic instruction mix and wait cycles of the whole-list kernel (per wave and op)
#   bash tools/pmc_instmix.sh [bench args]     default: the headline configuration
root=$(pwd)
out=$root/gpurun_out/instmix
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pass() { # tag, counters...
  local tag=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $out/$tag -- python3 $root/bench.py --steps 4 --warmup 1 --cpu-sites 0 --no-c4 $ARGS > /dev/null 2> $out/$tag.err
  python3 $root/tools/summarize_rocprof.py pmc $out/$tag $out/$tag.csv 2>/dev/null
  grep "${KERNEL:-k_dna_fused}" $out/$tag.csv | sed "s/^\"[^\"]*\"/fused/"
  rm -rf $out/$tag
}
ARGS="$*"
pass i1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS
pass i2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES
pass i3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass i4 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass i5 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
pass i6 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
pass i7 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH
