#!/usr/bin/env python3
"""developer tool: where the time of a pll_update_partials call with a NEW op list goes (BASELINE config 2's
partition, lists directed at different edges): the call that plans (first use of the list) against the
replays of the same list (kept plan), both from HIP events on the partition's stream and wall clock."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP

states = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sites = int(sys.argv[2]) if len(sys.argv) > 2 else (1_000_000 if states == 4 else 200_000)
lib = libpll_amd.load()
plan = W.balanced_tree(64, seed=42)
R = 4
cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
rates, freqs = (W.GTR_RATES, W.GTR_FREQS) if states == 4 else lib.aa_model("lg")
seqs = W.simulated_alignment(plan, sites, rates, freqs, cats, seed=42)
p = W.setup_partition(lib, plan, seqs, states, R, ATTRIB_PATTERN_TIP)
view = W.UnrootedView(plan)
rng = W.SplitMix64(777)
inner = [e for e in view.edges() if e[0] >= 64 and e[1] >= 64]
roots = [view.root] + [inner[rng.below(len(inner))] for _ in range(4)]
for _ in range(300):
    p.update_partials(plan.ops)
p.wait()
for r in roots:
    ops, edge = view.traversal(r)
    first, rep = [], []
    for trial in range(6):
        p.update_partials(plan.ops if r != view.root else view.traversal(roots[1])[0])  # another list in between
        p.wait()
        t0 = time.perf_counter(); p.timer_start(); p.update_partials(ops); ev = p.timer_stop_ms(); p.wait()
        first.append((ev, (time.perf_counter() - t0) * 1e3))
        for _ in range(3):
            t0 = time.perf_counter(); p.timer_start(); p.update_partials(ops); ev = p.timer_stop_ms(); p.wait()
            rep.append((ev, (time.perf_counter() - t0) * 1e3))
    f = np.median(np.array(first), axis=0); q = np.median(np.array(rep), axis=0)
    print("root %-10s first call: %.3f ms on the stream, %.3f ms wall   replay: %.3f / %.3f" % (r, f[0], f[1], q[0], q[1]))
