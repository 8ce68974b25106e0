#!/usr/bin/env python3
"""developer tool: CLV-update rate of the state counts that have no dedicated kernel
(2 = binary, 5 = DNA + gap, 61 = codons): tip CLVs, balanced 16-taxon tree, 4 rates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libpll_amd
from libpll_amd import workload as W
amd = libpll_amd.load()
taxa, rc = 16, 4
for states, sites in ((2, 1_000_000), (5, 1_000_000), (7, 500_000), (13, 300_000), (32, 100_000), (61, 60_000)):
    plan = W.balanced_tree(taxa, seed=42)
    rng = np.random.default_rng(7)
    p = amd.partition_create(taxa, taxa - 2, states, sites, 1, 2 * taxa - 3, rc, taxa - 2, 0)
    nsub = states * (states - 1) // 2
    p.set_subst_params(0, rng.uniform(0.5, 2.0, nsub))
    f = rng.uniform(0.5, 1.5, states); p.set_frequencies(0, f / f.sum())
    p.set_category_rates(amd.compute_gamma_cats(0.7, rc))
    for t in range(taxa):
        codes = rng.integers(0, states, sites)
        clv = np.zeros((sites, rc, states)); clv[np.arange(sites), :, codes] = 1.0
        p.set_tip_clv(t, clv.reshape(-1))
    p.update_prob_matrices([0] * rc, plan.matrix_indices, plan.branch_lengths)
    p.update_partials(plan.ops); p.wait()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps): p.update_partials(plan.ops)
    p.wait()
    dt = (time.perf_counter() - t0) / reps
    nops = len(plan.ops)
    bytes_op = sites * (3 * rc * states * 8 + 12)
    flop_op = sites * rc * states * (4 * states + 1)
    print("states %2d sites %8d: %.1f us/op, %.2f TB/s algorithmic, %.2f TFLOP/s f64, lnL %.4f"
          % (states, sites, dt / nops * 1e6, bytes_op / (dt / nops) / 1e12, flop_op / (dt / nops) / 1e12,
             p.compute_edge_loglikelihood(*plan.root_edge, [0] * rc)))
    e = plan.root_edge
    t0 = time.perf_counter()
    for _ in range(reps): p.compute_edge_loglikelihood(*e, [0] * rc)
    t_lnl = (time.perf_counter() - t0) / reps
    st = p.alloc_sumtable()
    t0 = time.perf_counter()
    for _ in range(reps): p.update_sumtable(e[0], e[2], e[1], e[3], [0] * rc, st)
    p.wait()
    t_sum = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): p.compute_likelihood_derivatives(e[1], e[3], 0.1, [0] * rc, st)
    t_der = (time.perf_counter() - t0) / reps
    print("          lnL call %.1f us (%.2f TB/s), sumtable %.1f us, derivative call %.1f us (%.2f TB/s)"
          % (t_lnl * 1e6, sites * 2 * rc * states * 8 / t_lnl / 1e12, t_sum * 1e6, t_der * 1e6,
             sites * rc * states * 8 / t_der / 1e12))
    p.destroy()
