#!/usr/bin/env python3
"""developer tool (round 6): the lists of traversal roots 0 and 1 of BASELINE config 3's partition under the tool
build's switches (PLLHIP_AF_EXP, build/afexp):  PLL_AMD_LIB=build/afexp/libpll_amd.so python3 tools/roots_tool_build_exp.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["PLLHIP_DEVELOPER"] = "1"
import numpy as np
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
taxa, sites, R = 64, 200000, 4
for mask in (0, 32, 64):
    os.environ["PLLHIP_AF_EXP"] = str(mask)
    lib = libpll_amd.load()
    plan = W.balanced_tree(taxa, seed=42)
    cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
    rates, freqs = lib.aa_model("lg")
    seqs = W.simulated_alignment(plan, sites, rates, freqs, cats, seed=42)
    p = W.setup_partition(lib, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
    view = W.UnrootedView(plan)
    rng = W.SplitMix64(777)
    inner = [e for e in view.edges() if e[0] >= taxa and e[1] >= taxa]
    roots = [view.root] + [inner[rng.below(len(inner))] for _ in range(4)]
    out = []
    for i in range(5):
        ops, edge = view.traversal(roots[i])
        for k in range(3):
            p.update_partials(ops)
        p.wait()
        ts = []
        for k in range(7):
            p.timer_start(); p.update_partials(ops); ts.append(p.timer_stop_ms() * 1e3)
        out.append("root %d %.0f" % (i, float(np.median(ts))))
    print("mask %2d:" % mask, "  ".join(out), flush=True)
    p.destroy()
