#!/usr/bin/env python3
"""developer tool: fixed per-call cost of the API entry points on a tiny partition"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
amd = libpll_amd.load()
for sites in (1000, 100000):
    plan = W.balanced_tree(64)
    seqs = W.random_alignment(64, sites, 4)
    p = W.setup_partition(amd, plan, seqs, 4, 4, ATTRIB_PATTERN_TIP)
    fi = [0] * 4
    e = plan.root_edge
    p.update_partials(plan.ops); p.wait()
    def timeit(f, n=300):
        f(); t = time.perf_counter()
        for _ in range(n): f()
        return (time.perf_counter() - t) / n * 1e6
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], fi, st)
    print("sites %d: update_partials(62 ops)+wait %.1f us | edge lnL %.1f us | derivatives %.1f us | "
          "update_sumtable+wait %.1f us | update_prob_matrices(126)+wait %.1f us | wait only %.1f us"
          % (sites, timeit(lambda: (p.update_partials(plan.ops), p.wait())),
             timeit(lambda: p.compute_edge_loglikelihood(*e, fi)),
             timeit(lambda: p.compute_likelihood_derivatives(e[1], e[3], 0.1, fi, st)),
             timeit(lambda: (p.update_sumtable(e[0], e[2], e[1], e[3], fi, st), p.wait())),
             timeit(lambda: (p.update_prob_matrices(fi, plan.matrix_indices, plan.branch_lengths), p.wait())),
             timeit(lambda: p.wait())))
    p.destroy()
