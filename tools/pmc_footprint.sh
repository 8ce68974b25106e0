#!/bin/bash
# counters of the whole-list kernel on the 128-taxon list at 500 k sites (8 GB of CLVs, fast)
# and 1 M sites (16 GB, slow): what differs?   bash tools/pmc_footprint.sh
root=$(pwd)
out=$root/gpurun_out/pmcfp2
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
pass() { # tag, sites, counters...
  local tag=$1 sites=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $out/${tag}_$sites -- python3 $root/bench.py --steps 4 --warmup 1 --cpu-sites 0 --taxa 128 --sites $sites > /dev/null 2> $out/${tag}_$sites.err
  python3 $root/tools/summarize_rocprof.py pmc $out/${tag}_$sites $out/${tag}_$sites.csv 2>/dev/null
  grep k_dna_fused $out/${tag}_$sites.csv | sed "s/^\"[^\"]*\"/$sites/"
  rm -rf $out/${tag}_$sites
}
for sites in 500000 1000000; do
  pass p1 $sites TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
  pass p2 $sites SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM
  pass p3 $sites TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
  pass p4 $sites TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
  pass p5 $sites TCC_BUSY_avr TCC_IB_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum
  pass p6 $sites MemUnitStalled VALUBusy SALUBusy
  pass p7 $sites TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum
done
