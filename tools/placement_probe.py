#!/usr/bin/env python3
"""developer tool (round 6): does the speed of a partition's stores depend on WHERE in device memory it lies?
(tools/clv_pad_sweep.sh on a fresh box: the first process stored BASELINE config 2's list at 5.7 TB/s and ran it in
1,483 us, every later process -- same code, same sizes -- at 6.1-7.3 TB/s and 1,385-1,400 us.)  One process: N partitions
of config 2 created one after the other and kept, each measured (list kernel by HIP events, bare stores of its list);
then all destroyed and three more created.  PLLHIP_PLACEMENT_TRIES=1 shows the places as the allocator hands them
out (what this tool was written for); the default shows what the library's search makes of them.
   python3 tools/placement_probe.py [partitions] [sites]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("PLLHIP_DEVELOPER", "1")
os.environ.setdefault("PLL_AMD_AUTO_MIRROR_MB", "0")
import torch
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP

n_parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
lib = libpll_amd.load()
lib.lib.pll_amd_set_device(0)
S, R, taxa = 4, 4, 64
plan = W.balanced_tree(taxa, seed=42)
cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
seqs = W.simulated_alignment(plan, sites, W.GTR_RATES, W.GTR_FREQS, cats, seed=42)
t_start = time.perf_counter()


def measure(p, label):
    for _ in range(20):
        p.update_partials(plan.ops)
    p.wait()
    us = []
    for _ in range(5):
        p.wait()
        p.timer_start()
        for _ in range(10):
            p.update_partials(plan.ops)
        us.append(p.timer_stop_ms() * 1e3 / 10)
    us.sort()
    # a pure read stream over the same memory: the edge log-likelihood (two CLVs of 128 MB), HIP events
    for _ in range(5):
        p.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
    p.wait()
    p.timer_start()
    for _ in range(20):
        p.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
    lnl_us = p.timer_stop_ms() * 1e3 / 20
    ms, nbytes = p.write_ceiling(plan.ops, 20)
    fill = sorted(p.arena_fill_bandwidth() for _ in range(5))
    p.update_partials(plan.ops)
    p.wait()
    free, total = torch.cuda.mem_get_info()
    label = "%s %s" % (label, p.placement())
    print("%-44s t = %5.1f s  list %8.1f us   lnL call %5.1f us   bare stores %8.1f us = %6.1f GB/s   contiguous fill %6.1f GB/s (%6.1f - %6.1f)   device memory in use %5.1f GB"
          % (label, time.perf_counter() - t_start, us[len(us) // 2], lnl_us, ms * 1e3, nbytes / ms / 1e6, fill[2], fill[0], fill[4], (total - free) / 1e9), flush=True)


parts = []
for i in range(n_parts):
    p = W.setup_partition(lib, plan, seqs, S, R, ATTRIB_PATTERN_TIP)
    parts.append(p)
    measure(p, "partition %d (the others kept)" % i)
print("-- every partition once more, in order")
for i, p in enumerate(parts):
    measure(p, "partition %d again" % i)
for p in parts:
    p.destroy()
parts = []
print("-- all destroyed; new ones")
for i in range(3):
    p = W.setup_partition(lib, plan, seqs, S, R, ATTRIB_PATTERN_TIP)
    parts.append(p)
    measure(p, "new partition %d" % i)
