#!/usr/bin/env python3
"""developer tool: the 20-state whole-list kernel AT SIZE -- 100,000 sites x 200 taxa, where a
workgroup walks enough tiles for its waves to drift apart (round 3's race between an inner-inner op and a run of
barrier-free lookups showed only there) -- on many random TREES over one alignment: two resident partitions, one on
the whole-list kernel (PLLHIP_FUSED=2), one on the per-level launches (PLLHIP_FUSED=0), are handed the same new op
list seed after seed (full traversal, the same list again = the kept plan, a partial traversal after a branch-length
change) and must agree bit for bit on lnL, per-site lnL and the scale buffers of the last ops; every 50th seed
on the CLVs of the last three ops too.
  python3 tools/soak_aa_fused_at_size.py [first seed] [count] [sites] [taxa] [states: 20 | 4]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ.pop("PLLHIP_AA_EXACT", None)
import numpy as np
import libpll_amd
from helpers import bits_equal, clv_err
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP


def run(first=0, count=200, sites=100_000, T=200, states=20, rate_scalers=False, quiet=False):
    """-> number of seeds with a mismatch (tests/test_gpu_soak_at_size.py runs this in the driver's GPU suite)"""
    from libpll_amd.pllapi import ATTRIB_RATE_SCALERS
    amd = libpll_amd.load()
    R = 4
    rates, freqs = amd.aa_model("lg") if states == 20 else (W.GTR_RATES, W.GTR_FREQS)
    plan0 = W.random_tree(T, seed=42)
    seqs = W.simulated_alignment(plan0, sites, rates, freqs, amd.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    attrs = ATTRIB_PATTERN_TIP | (ATTRIB_RATE_SCALERS if rate_scalers else 0)
    parts = {}
    saved = os.environ.get("PLLHIP_FUSED")
    for fused in ("2", "0"):
        os.environ["PLLHIP_FUSED"] = fused
        parts[fused] = W.setup_partition(amd, plan0, seqs, states, R, attrs)
    if saved is None:
        os.environ.pop("PLLHIP_FUSED", None)
    else:
        os.environ["PLLHIP_FUSED"] = saved
    loose = states == 20 and os.environ.get("PLLHIP_AA_TI_MFMA", "1") != "0"
    bad, t0 = 0, time.time()
    for seed in range(first, first + count):
        plan = W.random_tree(T, seed=1000 + seed)
        rng = np.random.default_rng(seed)
        n = int(rng.integers(1, len(plan.ops)))                       # a partial traversal after a branch-length change
        t_new = float(rng.uniform(0.01, 1.2))
        out = {}
        for fused, p in parts.items():
            p.update_prob_matrices([0] * R, plan.matrix_indices, plan.branch_lengths)
            p.update_partials(plan.ops)
            a = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
            p.update_partials(plan.ops)                               # the kept plan
            b = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
            p.update_prob_matrices([0] * R, [int(plan.ops[-n]["child1_matrix_index"])], [t_new])
            p.update_partials(plan.ops[-n:])
            c = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
            scs = [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops[-3:]]
            clvs = [p.get_clv(int(op["parent_clv_index"])) for op in plan.ops[-3:]] if seed % 50 == 0 else []
            out[fused] = (a, b, c, scs, clvs)
        x, y = out["2"], out["0"]
        if loose:
            # 20 states, default path (round 6): the whole-list kernel's tip-inner mat-vecs run on the matrix cores --
            # scale buffers bit for bit (the scaling certificate), everything else to rounding; the kept plan's
            # evaluation still repeats the first one's bits
            ok = all(abs(x[k][0] - y[k][0]) <= 1e-12 * abs(y[k][0]) and np.abs(x[k][1] - y[k][1]).max() <= 1e-9 for k in range(3)) and \
                x[0][0] == x[1][0] and bits_equal(x[0][1], x[1][1]) and \
                all((s == t).all() for s, t in zip(x[3], y[3])) and all(clv_err(s, t) <= 1e-13 for s, t in zip(x[4], y[4]))
        else:
            ok = all(x[k][0] == y[k][0] and bits_equal(x[k][1], y[k][1]) for k in range(3)) and \
                x[0][0] == x[1][0] and bits_equal(x[0][1], x[1][1]) and \
                all((s == t).all() for s, t in zip(x[3], y[3])) and all(bits_equal(s, t) for s, t in zip(x[4], y[4]))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, [x[k][0] for k in range(3)], [y[k][0] for k in range(3)], flush=True)
        if (seed - first) % 500 == 499 and not quiet:
            print("  ... %d seeds, %d mismatches, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
    cert = parts["2"].scaling_certificate() if states == 20 else None
    for p in parts.values():
        p.destroy()
    print("soak_aa_fused_at_size: %d seeds from %d at %d sites x %d taxa, %d states%s, %d mismatches, %.0f s"
          % (count, first, sites, T, states, ", per-rate scale buffers" if rate_scalers else "", bad, time.time() - t0))
    if cert is not None:
        print("  scaling certificate of the whole-list partition: %s%s" % (cert, "" if loose else " (tip-inner mat-vecs in the reference's order)"))
        if cert["uncertified"]:
            bad += 1
    return bad


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    sys.exit(1 if run(*a) else 0)
