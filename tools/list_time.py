#!/usr/bin/env python3
"""developer tool: HIP-event time of pll_update_partials on a bench.py shape, nothing else -- works with older builds
of the library too (PLL_AMD_LIB=...), which bench.py's newer legs do not.
   python3 tools/list_time.py [states] [sites] [taxa] [tree: balanced|random|caterpillar] [rate_scalers 0|1]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("PLL_AMD_AUTO_MIRROR_MB", "0")
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS

states = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
taxa = int(sys.argv[3]) if len(sys.argv) > 3 else 64
tree = sys.argv[4] if len(sys.argv) > 4 else "balanced"
rs = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lib = libpll_amd.load()
plan = {"balanced": W.balanced_tree, "random": W.random_tree, "caterpillar": W.caterpillar_tree}[tree](taxa, seed=42)
R = 4
cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
rates, freqs = (W.GTR_RATES, W.GTR_FREQS) if states == 4 else lib.aa_model("lg")
seqs = W.simulated_alignment(plan, sites, rates, freqs, cats, seed=42)
p = W.setup_partition(lib, plan, seqs, states, R, ATTRIB_PATTERN_TIP | (ATTRIB_RATE_SCALERS if rs else 0))
for _ in range(30):
    p.update_partials(plan.ops)
p.wait()
us = []
for _ in range(9):
    p.wait()
    p.timer_start()
    for _ in range(10):
        p.update_partials(plan.ops)
    us.append(p.timer_stop_ms() * 1e3 / 10)
us.sort()
lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
print("%-40s %d states %7d sites %3d taxa %-11s pll_update_partials %8.1f us (min %8.1f, max %8.1f)  lnL %.6f"
      % (os.path.basename(os.environ.get("PLL_AMD_LIB") or "this build"), states, sites, taxa, tree, us[4], us[0], us[-1], lnl))
