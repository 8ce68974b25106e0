#!/usr/bin/env python3
"""developer tool: short op lists (the dirty path after one branch changed) on a 64-taxon x N-site partition,
whole-list kernel vs per-level launches: PLLHIP_FUSED=2|0 python tools/partial_traversal_timing.py N"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
amd = libpll_amd.load()
sites = int(sys.argv[1])
plan = W.balanced_tree(64, seed=42)
seqs = W.simulated_alignment(plan, sites, W.GTR_RATES, W.GTR_FREQS, amd.compute_gamma_cats(W.GAMMA_ALPHA, 4), seed=42)
p = W.setup_partition(amd, plan, seqs, 4, 4, ATTRIB_PATTERN_TIP)
p.update_partials(plan.ops); p.wait()
# (balanced_tree lists its ops level by level: 0..31 tip-tip, 32..47, 48..55, 56..59, 60, 61)
sub8 = [0, 1, 2, 3, 32, 33, 48]
sub16 = list(range(8)) + [32, 33, 34, 35, 48, 49, 56]
sub32 = list(range(16)) + list(range(32, 40)) + [48, 49, 50, 51, 56, 57, 60]
lists = ([60], [56, 60], [48, 56, 60], [0, 32, 48, 56, 60], sub8, sub16, sub32, list(range(32)))
if len(sys.argv) > 2:  # one list only, e.g. 56,60 (for a profiler run)
    lists = ([int(x) for x in sys.argv[2].split(",")],)
for idx in lists:
    ops = plan.ops[idx]
    p.update_partials(ops); p.wait()
    t = time.perf_counter()
    for _ in range(20): p.update_partials(ops)
    p.wait()
    print(os.environ.get("PLLHIP_FUSED"), sites, len(idx), "ops: %.1f us" % ((time.perf_counter() - t) / 20 * 1e6))
