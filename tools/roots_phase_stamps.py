#!/usr/bin/env python3
"""developer tool (round 6): the phase stamps of the timing build (tools/aa_fused_timing.sh build) for the lists of
traversal roots 0 and 1 of BASELINE config 3's partition:  PLL_AMD_LIB=build/aftiming/libpll_amd.so python3 tools/roots_phase_stamps.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["PLLHIP_DEVELOPER"] = "1"
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
lib = libpll_amd.load()
taxa, sites, R = 64, 200000, 4
plan = W.balanced_tree(taxa, seed=42)
cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
rates, freqs = lib.aa_model("lg")
seqs = W.simulated_alignment(plan, sites, rates, freqs, cats, seed=42)
p = W.setup_partition(lib, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
view = W.UnrootedView(plan)
rng = W.SplitMix64(777)
inner = [e for e in view.edges() if e[0] >= taxa and e[1] >= taxa]
roots = [view.root] + [inner[rng.below(len(inner))] for _ in range(4)]
for i in (0, 1):
    ops, edge = view.traversal(roots[i])
    for k in range(3):
        p.update_partials(ops); p.wait()
    print("==== root %d %s %s" % (i, roots[i], p.list_kinds()), flush=True)
    p.timer_start()
    p.update_partials(ops)
    print("     %.0f us" % (p.timer_stop_ms() * 1e3), flush=True)
