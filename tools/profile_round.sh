#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash tools/profile_round.sh <tag>
# Takes, for each BASELINE workload, the plain bench line, the rocprofv3 kernel statistics of
# the same workload (with --no-vary: the varying-lists leg launches the same kernels on short and re-rooted lists and
# (and --no-c4: the N = 1 line of the default workload also evaluates config 4 whole) would dilute the per-kernel averages the bench line's avg_launch_us is to be compared with) and the HBM traffic counters (FETCH_SIZE and WRITE_SIZE in passes of their
# own: no trace domains next to --pmc), condenses them with tools/summarize_rocprof.py into
# gpurun_out/<tag>/summary/ (what gets copied into profiles/) and writes the index bench.py reads
# `roofline.traffic` from.
set -u
tag=${1:-r5}
root=$(pwd)
out=$root/gpurun_out/$tag
sum=$out/summary
mkdir -p "$sum"
export TMPDIR=/tmp
cd /tmp

run() { # name, [ENV=VAL ...] -- bench args...
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  for e in "${envs[@]}"; do export "$e"; done
  if [ "$phase" = pmc ]; then
    # first phase: the HBM counters (their index is what the later bench lines look roofline.traffic up in)
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch_$name" -- \
        python3 "$root/bench.py" --steps 5 --warmup 1 --cpu-sites 0 --no-vary --no-c4 "$@" > "$out/pmcpass_$name.json" 2> "$out/fetch_$name.err"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write_$name" -- \
        python3 "$root/bench.py" --steps 5 --warmup 1 --cpu-sites 0 --no-vary --no-c4 "$@" > /dev/null 2> "$out/write_$name.err"
    python3 "$root/tools/summarize_rocprof.py" hbm "$out/fetch_$name" "$out/write_$name" "$sum/${tag}_pmc_hbm_traffic_$name.csv" \
        "${envs[*]} python3 bench.py --steps 5 --warmup 1 --cpu-sites 0 --no-vary --no-c4 $*"
    rm -rf "$out/fetch_$name" "$out/write_$name"
  else
    # second phase: the kernel statistics of the same workload, then the plain line
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$name" -- \
        python3 "$root/bench.py" --steps 20 --warmup 3 --cpu-sites 0 --no-vary --no-c4 "$@" > "$sum/${tag}_bench_${name}_under_rocprof.json" 2> "$out/trace_$name.err"
    python3 "$root/tools/summarize_rocprof.py" stats "$out/trace_$name" "$sum/${tag}_bench_${name}_kernel_stats.csv"
    rm -rf "$out/trace_$name"
    (cd "$root" && python3 bench.py "$@" > "$sum/${tag}_bench_$name.json" 2> "$out/bench_$name.err")
  fi
  for e in "${envs[@]}"; do unset "${e%%=*}"; done
}

workloads() {
  run c2 --
  run c2_per_level PLLHIP_FUSED=0 --
  run c3 -- --states 20 --sites 200000
  run c3_per_level PLLHIP_FUSED=0 -- --states 20 --sites 200000
  run c3_random_200 -- --states 20 --sites 100000 --taxa 200 --tree random
  run c4_shard -- --taxa 128
  run c5_shape -- --sites 500000 --taxa 200 --tree random --newton 5
  run c2_tip_clv -- --tip-clv
}
phase=pmc
workloads
# BASELINE config 4 WHOLE on this GPU (133 GB): the HBM counters of the one-GPU point of the strong-scaling curve
# (the program directly behind "--")
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$out/${ctr}_c4_whole" -- \
      python3 "$root/bench.py" --total-sites 8000000 --taxa 128 --steps 3 --warmup 1 --cpu-sites 0 --no-vary --no-c4 > "$out/pmcpass_c4_whole_$ctr.json" 2> "$out/${ctr}_c4_whole.err"
done
python3 "$root/tools/summarize_rocprof.py" hbm "$out/FETCH_SIZE_c4_whole" "$out/WRITE_SIZE_c4_whole" "$sum/${tag}_pmc_hbm_traffic_c4_whole.csv" \
    "python3 bench.py --total-sites 8000000 --taxa 128 --steps 3 --warmup 1 --cpu-sites 0 --no-vary --no-c4"
rm -rf "$out/FETCH_SIZE_c4_whole" "$out/WRITE_SIZE_c4_whole"
for shape in "c3_random_200 --states 20 --sites 100000 --taxa 200 --tree random"; do
  set -- $shape; nm=$1; shift
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/mfma_$nm" -- \
      python3 "$root/bench.py" --steps 5 --warmup 1 --cpu-sites 0 --no-vary "$@" > /dev/null 2> "$out/mfma_$nm.err"
  python3 "$root/tools/summarize_rocprof.py" pmc "$out/mfma_$nm" "$sum/${tag}_pmc_mfma_$nm.csv" "python3 bench.py --steps 5 --warmup 1 --cpu-sites 0 --no-vary $*"
  rm -rf "$out/mfma_$nm"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/mfma_c3" -- \
    python3 "$root/bench.py" --steps 5 --warmup 1 --cpu-sites 0 --no-vary --states 20 --sites 200000 > /dev/null 2> "$out/mfma_c3.err"
python3 "$root/tools/summarize_rocprof.py" pmc "$out/mfma_c3" "$sum/${tag}_pmc_mfma_c3.csv" "python3 bench.py --steps 5 --warmup 1 --cpu-sites 0 --no-vary --states 20 --sites 200000"
rm -rf "$out/mfma_c3"
cat > "$sum/pmc_spec.json" <<EOF
[
 {"csv": "${tag}_pmc_hbm_traffic_c2.csv", "kernel_match": "k_dna_fused", "kernel_class": "whole-list",
  "workload": {"states": 4, "rate_cats": 4, "sites": 1000000, "taxa": 64, "tree": "balanced", "tip_clv": false, "rate_scalers": false}},
 {"csv": "${tag}_pmc_hbm_traffic_c4_shard.csv", "kernel_match": "k_dna_fused", "kernel_class": "whole-list",
  "workload": {"states": 4, "rate_cats": 4, "sites": 1000000, "taxa": 128, "tree": "balanced", "tip_clv": false, "rate_scalers": false}},
 {"csv": "${tag}_pmc_hbm_traffic_c4_whole.csv", "kernel_match": "k_dna_fused", "kernel_class": "whole-list",
  "workload": {"states": 4, "rate_cats": 4, "sites": 8000000, "taxa": 128, "tree": "balanced", "tip_clv": false, "rate_scalers": false}},
 {"csv": "${tag}_pmc_hbm_traffic_c5_shape.csv", "kernel_match": "k_dna_fused", "kernel_class": "whole-list",
  "workload": {"states": 4, "rate_cats": 4, "sites": 500000, "taxa": 200, "tree": "random", "tip_clv": false, "rate_scalers": false}},
 {"csv": "${tag}_pmc_hbm_traffic_c2_tip_clv.csv", "kernel_match": "k_dna_fused", "kernel_class": "whole-list",
  "workload": {"states": 4, "rate_cats": 4, "sites": 1000000, "taxa": 64, "tree": "balanced", "tip_clv": true, "rate_scalers": false}},
 {"csv": "${tag}_pmc_hbm_traffic_c2_per_level.csv", "kernel_match": "k_dna_partials<4, 1, true, 0", "kernel_class": "inner-inner",
  "bench_json": "$out/pmcpass_c2_per_level.json",
  "workload": {"states": 4, "rate_cats": 4, "sites": 1000000, "taxa": 64, "tree": "balanced", "tip_clv": false, "rate_scalers": false}},
 {"csv": "${tag}_pmc_hbm_traffic_c3.csv", "kernel_match": "k_aa_fused", "kernel_class": "whole-list", "sum_call": true, "mfma_csv": "${tag}_pmc_mfma_c3.csv",
  "exclude": ["k_lnl", "k_update_pmatrix", "fillBuffer", "copyBuffer", "k_final_sum"],
  "workload": {"states": 20, "rate_cats": 4, "sites": 200000, "taxa": 64, "tree": "balanced", "tip_clv": false, "rate_scalers": false}},
 {"csv": "${tag}_pmc_hbm_traffic_c3_random_200.csv", "kernel_match": "k_aa_fused", "kernel_class": "whole-list", "sum_call": true, "mfma_csv": "${tag}_pmc_mfma_c3_random_200.csv",
  "exclude": ["k_lnl", "k_update_pmatrix", "fillBuffer", "copyBuffer", "k_final_sum"],
  "workload": {"states": 20, "rate_cats": 4, "sites": 100000, "taxa": 200, "tree": "random", "tip_clv": false, "rate_scalers": false}},
 {"csv": "${tag}_pmc_hbm_traffic_c3_per_level.csv", "kernel_match": "k_aa_ii_mfma<4, 1", "kernel_class": "inner-inner",
  "bench_json": "$out/pmcpass_c3_per_level.json",
  "workload": {"states": 20, "rate_cats": 4, "sites": 200000, "taxa": 64, "tree": "balanced", "tip_clv": false, "rate_scalers": false}}
]
EOF
python3 "$root/tools/summarize_rocprof.py" index "$sum/pmc_spec.json" "$sum/pmc_traffic.json"
# the bench lines that follow read roofline.traffic from the index of THIS run
cp "$sum/pmc_traffic.json" "$root/profiles/pmc_traffic.json"
phase=lines
workloads
# BASELINE config 4 on ONE GPU (133 GB): the strong-scaling reference point
(cd "$root" && python3 bench.py --total-sites 8000000 --taxa 128 --cpu-sites 0 --steps 10 > "$sum/${tag}_bench_c4_one_gpu.json" 2> "$out/bench_c4_one_gpu.err")
# the N > 1 paths on this one GPU: one process, a partition the library shards over "four devices" (ordinal 0 four
# times) -- lnL checked against the reference and against the one-GPU evaluation of the same alignment --, and the
# RCCL path (communicator, lnL all-reduce) on one rank
(cd "$root" && python3 bench.py --gpus 4 --in-process --devices 0,0,0,0 --sites 250000 --steps 10 > "$sum/${tag}_bench_in_process_4_shards.json" 2> "$out/bench_inproc4.err")
(cd "$root" && python3 bench.py --gpus 8 --in-process --devices 0,0,0,0,0,0,0,0 --total-sites 1000000 --steps 10 --cpu-sites 0 --no-vary > "$sum/${tag}_bench_in_process_8_shards_of_c2.json" 2> "$out/bench_inproc8.err")
(cd "$root" && python3 bench.py --force-comm --no-c4 --steps 10 > "$sum/${tag}_bench_force_comm.json" 2> "$out/bench_forcecomm.err")
# the sizes an 8-way split of the BASELINE configs leaves a GPU (round 5: segments)
(cd "$root" && bash tools/size_sweep2.sh > "$sum/${tag}_size_sweep.txt" 2>&1)
# site repeats on the C5 shape
(cd "$root" && python3 bench.py --sites 500000 --taxa 200 --tree random --site-repeats --cpu-sites 0 > "$sum/${tag}_bench_c5_shape_site_repeats.json" 2> "$out/bench_c5rep.err")
# round 6: per-rate scale buffers on the 20-state whole-list kernel, a ladder (every op but one tip-inner), and the
# result-returning calls from C (the Newton loop and the three-op step of the reference's examples)
(cd "$root" && python3 bench.py --states 20 --sites 200000 --rate-scalers --cpu-sites 0 --no-vary > "$sum/${tag}_bench_c3_rate_scalers.json" 2> "$out/bench_c3_rs.err")
(cd "$root" && python3 bench.py --states 20 --sites 100000 --taxa 100 --tree caterpillar --cpu-sites 0 --no-vary > "$sum/${tag}_bench_aa_ladder_100.json" 2> "$out/bench_ladder.err")
(cd "$root" && { for cfg in "4 500000" "4 1000000" "20 200000"; do echo "== tools/newton_floor.bin $cfg"; tools/newton_floor.bin $cfg; done; for cfg in "4 2000" "4 12000" "20 2000" "20 12000"; do echo "== PLL_AMD_AUTO_MIRROR_MB=0 tools/step_floor.bin $cfg"; PLL_AMD_AUTO_MIRROR_MB=0 tools/step_floor.bin $cfg; done; } > "$sum/${tag}_result_calls_from_c.txt" 2>&1)
ls -la "$sum"
