#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the GENUINE reference.

Run in the dev container (needs oracle/_ref/libpll_ref.so, built from
/root/reference by `make -C oracle ref`):

    python tests/golden/make_golden.py

Each fixture is plain data: the inputs of one evaluation (tip sequences, op
list, branch lengths, model parameters, attribute word) and the outputs the
reference's AVX2-flag path produced for them (category rates, eigen system,
P-matrices, inner CLVs, scale buffers, per-site lnL, lnL, sumtable, d/dd).
The fixtures travel to the GPU box, where /root/reference does not exist.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from helpers import make_case, odd_state_case, build_partition, invariant_of  # noqa: E402
from libpll_amd.pllapi import (PllLibrary, ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS,  # noqa: E402
                               ATTRIB_ARCH_AVX2, ATTRIB_ARCH_CPU)

ref = PllLibrary(os.path.join(ROOT, "oracle", "_ref", "libpll_ref.so"))

# name: (states, shape, tips, sites, attrs, pinv, kwargs, which CLVs to keep)
SPECS = {
    "dna_balanced16_tipclv_site": (4, "balanced", 16, 96, 0, 0.0, {}, "all"),
    "dna_balanced16_pattern_site": (4, "balanced", 16, 96, ATTRIB_PATTERN_TIP, 0.0, {}, "all"),
    "dna_random23_pattern_rate": (4, "random", 23, 77, ATTRIB_PATTERN_TIP | ATTRIB_RATE_SCALERS, 0.0, {}, "all"),
    "dna_random23_pattern_pinv": (4, "random", 23, 77, ATTRIB_PATTERN_TIP, 0.25, {"columns_constant": 3}, "all"),
    "dna_caterpillar700_site": (4, "caterpillar", 700, 8, ATTRIB_PATTERN_TIP, 0.0,
                                dict(alpha=0.5, branch=0.5, weights=False, ambiguity=False, gap_frac=0.0), "last4"),
    "dna_caterpillar700_rate": (4, "caterpillar", 700, 8, ATTRIB_PATTERN_TIP | ATTRIB_RATE_SCALERS, 0.0,
                                dict(alpha=0.5, branch=0.5, weights=False, ambiguity=False, gap_frac=0.0), "last4"),
    "aa_balanced8_pattern_site": (20, "balanced", 8, 40, ATTRIB_PATTERN_TIP, 0.0, {}, "all"),
    "aa_random11_tipclv_rate": (20, "random", 11, 31, ATTRIB_RATE_SCALERS, 0.0, {}, "all"),
    "aa_caterpillar400_site": (20, "caterpillar", 400, 8, ATTRIB_PATTERN_TIP, 0.0,
                               dict(alpha=0.5, branch=0.5, weights=False, ambiguity=False, gap_frac=0.0), "last4"),
    "odd5_random9": (5, "random", 9, 30, 0, 0.0, {}, "all"),
}


def generate(name, spec):
    states, shape, tips, sites, attrs, pinv, kw, keep = spec
    kw = dict(kw)
    const_every = kw.pop("columns_constant", 0)
    if states in (4, 20):
        case = make_case(states, shape, tips, sites, seed=tips * 3 + sites, **kw)
        arch = ATTRIB_ARCH_AVX2
    else:
        case = odd_state_case(states, tips, sites)
        arch = ATTRIB_ARCH_CPU
    if states == 20:
        case["rates"], case["freqs"] = ref.aa_model("lg")
    if const_every:
        seqs = [bytearray(s) for s in case["seqs"]]
        for col in range(0, sites, const_every):
            for s in seqs:
                s[col] = seqs[0][col] if chr(seqs[0][col]) in "ACGT" else ord("A")
        case["seqs"] = [bytes(s) for s in seqs]
    p = build_partition(ref, case, attrs | arch, pinv=pinv)
    plan = case["plan"]
    R = case["rate_cats"]
    vals, vecs, inv = p.get_eigen(0)
    out = dict(
        states=states, rate_cats=R, sites=sites, tips=tips, attributes=attrs, pinv=pinv,
        alpha=case["alpha"], subst_params=np.asarray(case["rates"], dtype=np.float64),
        freqs=np.asarray(case["freqs"], dtype=np.float64),
        seqs=np.stack([np.frombuffer(s, dtype=np.uint8) for s in case["seqs"]]),
        cmap=np.zeros(0, dtype=np.uint32) if case["cmap"] is None else case["cmap"],
        pattern_weights=np.zeros(0, dtype=np.uint32) if case["pw"] is None else case["pw"],
        ops=plan.ops, matrix_indices=plan.matrix_indices, branch_lengths=plan.branch_lengths,
        root_edge=np.array(plan.root_edge, dtype=np.int64),
        # ---- reference outputs ----
        cat_rates=ref.compute_gamma_cats(case["alpha"], R),
        eigenvals=vals, eigenvecs=vecs, inv_eigenvecs=inv,
        pmatrices=np.stack([p.get_pmatrix(int(m)) for m in plan.matrix_indices]),
    )
    inv_arr = invariant_of(p)
    out["invariant"] = np.zeros(0, dtype=np.int32) if inv_arr is None else inv_arr
    p.update_partials(plan.ops)
    ops = plan.ops if keep == "all" else plan.ops[-4:]
    out["kept_nodes"] = ops["parent_clv_index"].astype(np.int64)
    out["clvs"] = np.stack([p.get_clv(int(n)) for n in ops["parent_clv_index"]])
    out["scalers"] = np.stack([p.get_scaler(int(i)) for i in plan.ops["parent_scaler_index"]])
    lnl, persite = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
    out["lnl"] = lnl
    out["persite_lnl"] = persite
    e = plan.root_edge
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    out["sumtable"] = p.get_sumtable(st).copy()
    ts = np.array([0.01, 0.13, 0.9, 4.0])
    out["deriv_t"] = ts
    out["deriv"] = np.array([p.compute_likelihood_derivatives(e[1], e[3], float(t), [0] * R, st)
                             for t in ts])
    p.destroy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    return path, lnl, int(out["scalers"].max())


if __name__ == "__main__":
    for name, spec in SPECS.items():
        path, lnl, smax = generate(name, spec)
        print("%-34s lnL %.10f  max scaler %d  %6.1f KB" % (name, lnl, smax,
                                                          os.path.getsize(path) / 1024))
