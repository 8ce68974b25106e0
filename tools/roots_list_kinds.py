#!/usr/bin/env python3
"""developer tool (round 6): what the planner makes of the five traversal roots' lists of tools/replay_probe.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["PLLHIP_DEVELOPER"] = "1"
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
lib = libpll_amd.load()
taxa, sites, R = 64, 20000, 4
plan = W.balanced_tree(taxa, seed=42)
cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
rates, freqs = lib.aa_model("lg")
seqs = W.simulated_alignment(plan, sites, rates, freqs, cats, seed=42)
os.environ["PLLHIP_FUSED"] = "2"
p = W.setup_partition(lib, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
view = W.UnrootedView(plan)
rng = W.SplitMix64(777)
inner = [e for e in view.edges() if e[0] >= taxa and e[1] >= taxa]
roots = [view.root] + [inner[rng.below(len(inner))] for _ in range(4)]
for r in roots:
    ops, edge = view.traversal(r)
    p.update_partials(ops)
    p.wait()
    print(r, p.list_kinds())
