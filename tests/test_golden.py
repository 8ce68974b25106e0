"""Golden vectors (tests/golden/*.npz, produced from the genuine reference by
tests/golden/make_golden.py) against (a) the oracle on CPU and (b) the HIP path
through the C-ABI on the GPU.  Bit-exact for P-matrices, CLVs, scalers and
per-site lnL of 4- and 20-state data; lnL to 1e-12 relative (the sum over sites
is a tree on the GPU, sequential in the reference)."""
import glob
import os

import numpy as np
import pytest

from helpers import bits_equal, clv_ok, rel_err, sumtable_err
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
from oracle_api import OracleRun

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
IDS = [os.path.basename(f)[:-4] for f in FIXTURES]


def load(path):
    g = dict(np.load(path))
    plan = W.TreePlan(int(g["tips"]), g["ops"], g["matrix_indices"], g["branch_lengths"],
                      tuple(int(x) for x in g["root_edge"]))
    g["plan"] = plan
    for k in ("states", "rate_cats", "sites", "tips", "attributes"):
        g[k] = int(g[k])
    g["pinv"] = float(g["pinv"])
    g["alpha"] = float(g["alpha"])
    g["pw"] = g["pattern_weights"] if g["pattern_weights"].size else None
    return g


def test_fixtures_present():
    assert len(FIXTURES) >= 10


def charmap(lib_maps, g):
    if g["cmap"].size:
        return g["cmap"]
    return lib_maps("nt" if g["states"] == 4 else "aa")


def host_tip_encoding(g, cmap):
    """tip codes + tipmap the way create_charmap assigns them (pll.c:305-325)."""
    S = g["states"]
    seqs = g["seqs"]
    if S == 4:
        return cmap[seqs].astype(np.uint8), np.zeros(256, dtype=np.uint32)
    tipmap = np.zeros(256, dtype=np.uint32)
    code_of = {}
    for ch in range(256):
        m = int(cmap[ch])
        if m and m not in code_of:
            code_of[m] = len(code_of)
            tipmap[code_of[m]] = m
    lut = np.zeros(256, dtype=np.uint8)
    for ch in range(256):
        if cmap[ch]:
            lut[ch] = code_of[int(cmap[ch])]
    return lut[seqs], tipmap


def check_outputs(g, pmats, clv_of, scaler_of, lnl, persite, sumtable, derivs, exact_lnl, mfma_lnl=False, clv_exact=True):
    """mfma_lnl: the 20-state DEFAULT path -- P-matrices, every CLV and every scaler count still bit for bit
    (inner-inner ops on the matrix cores in the reference's order, tip-inner ops on the vector unit); the
    edge-lnL and sumtable kernels add a row's products in one fused chain: last-bit differences there."""
    S = g["states"]
    exact = S in (4, 20, 5)
    for i in range(len(g["matrix_indices"])):
        assert bits_equal(pmats[i], g["pmatrices"][i]), "P-matrix %d" % i
    for i, node in enumerate(g["kept_nodes"]):
        assert clv_ok(clv_of(int(node)), g["clvs"][i], clv_exact), "CLV of node %d" % node
    for i, op in enumerate(g["plan"].ops):
        if int(op["parent_scaler_index"]) >= 0:
            assert (scaler_of(int(op["parent_scaler_index"])) == g["scalers"][i]).all(), "scaler %d" % i
    if exact:
        assert rel_err(persite, g["persite_lnl"]) < (1e-11 if mfma_lnl else 1e-13)
    if exact_lnl:
        assert lnl == float(g["lnl"])
    else:
        assert abs(lnl - float(g["lnl"])) <= (1e-11 if mfma_lnl else 1e-12) * abs(float(g["lnl"]))
    assert sumtable_err(sumtable, g["sumtable"]) < (1e-10 if mfma_lnl else 1e-12)
    assert rel_err(derivs, g["deriv"]) < 1e-10


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_oracle_matches_golden(orc, amd, path):
    g = load(path)
    S, R = g["states"], g["rate_cats"]
    model = dict(states=S, rate_cats=R, rates=g["cat_rates"], rate_weights=np.full(R, 1.0 / R),
                 eigenvals=g["eigenvals"], eigenvecs=g["eigenvecs"],
                 inv_eigenvecs=g["inv_eigenvecs"], freqs=g["freqs"], pinv=g["pinv"])
    cmap = charmap(amd.map, g)
    inv = g["invariant"] if g["invariant"].size else None
    if g["attributes"] & ATTRIB_PATTERN_TIP:
        codes, tipmap = host_tip_encoding(g, cmap)
        o = OracleRun(orc, model, g["plan"], g["attributes"], tipcodes=codes, tipmap=tipmap,
                      pattern_weights=g["pw"], invariant=inv)
    else:
        masks = cmap[g["seqs"]]
        clvs = ((masks[:, :, None] >> np.arange(S)) & 1)[:, :, None, :].repeat(R, axis=2)
        o = OracleRun(orc, model, g["plan"], g["attributes"], tipclvs=clvs.astype(np.float64),
                      pattern_weights=g["pw"], invariant=inv)
    o.update_partials()
    e = g["plan"].root_edge
    lnl, ps = o.edge_loglikelihood(*e, persite=True)
    st = o.sumtable(e[0], e[2], e[1], e[3])
    d = np.array([o.derivatives(st, float(t)) for t in g["deriv_t"]])
    check_outputs(g, [o.pmat[int(m)] for m in g["matrix_indices"]], lambda n: o.clv[n],
                  lambda i: o.scalers[i], lnl, ps, st, d, exact_lnl=True)


@pytest.mark.gpu
@pytest.mark.parametrize("aa_path", ["default", "reference-order", "vector-kernels"])
@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_hip_matches_golden(gpu, path, monkeypatch, dna_path, aa_path):
    """The product, driven exactly like a reference client, reproduces the
    reference's stored outputs -- 20-state fixtures on the DEFAULT path (matrix cores: every scaler count bit for bit,
    every CLV too except, round 6, below tip-inner ops of the whole-list kernel, whose mat-vec runs on the matrix cores:
    1e-13), with PLLHIP_AA_TI_MFMA=0 (every CLV bit for bit, as until round 5) and on the all-vector kernels
    (PLLHIP_AA_EXACT=1: per-site lnL bit for bit as well)."""
    g = load(path)
    S, R, plan = g["states"], g["rate_cats"], g["plan"]
    if S != 20 and aa_path != "default":
        pytest.skip("PLLHIP_AA_EXACT only affects 20-state kernels")
    monkeypatch.setenv("PLLHIP_AA_EXACT", "1" if aa_path == "vector-kernels" else "0")
    monkeypatch.setenv("PLLHIP_AA_TI_MFMA", "0" if aa_path == "reference-order" else "1")
    monkeypatch.setenv("PLLHIP_AA_CHERRY", "2")  # (the table-lookup ops also for these small partitions)
    p = gpu.partition_create(plan.tips, plan.clv_buffers, S, g["sites"], 1, plan.prob_matrices, R,
                             plan.scale_buffers, g["attributes"])
    p.set_frequencies(0, g["freqs"])
    p.set_subst_params(0, g["subst_params"])
    rates = gpu.compute_gamma_cats(g["alpha"], R)
    assert bits_equal(rates, g["cat_rates"])
    p.set_category_rates(rates)
    cmap = charmap(gpu.map, g)
    for i in range(plan.tips):
        p.set_tip_states(i, cmap, g["seqs"][i].tobytes())
    if g["pw"] is not None:
        p.set_pattern_weights(g["pw"])
    if g["pinv"] > 0:
        p.update_invariant_sites_proportion(0, g["pinv"])
    p.update_prob_matrices([0] * R, plan.matrix_indices, plan.branch_lengths)
    vals, vecs, inv = p.get_eigen(0)
    assert bits_equal(vals, g["eigenvals"]) and bits_equal(vecs, g["eigenvecs"])
    assert bits_equal(inv, g["inv_eigenvecs"])
    p.update_partials(plan.ops)
    e = plan.root_edge
    lnl, ps = p.compute_edge_loglikelihood(*e, [0] * R, persite=True)
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    d = np.array([p.compute_likelihood_derivatives(e[1], e[3], float(t), [0] * R, st)
                  for t in g["deriv_t"]])
    check_outputs(g, [p.get_pmatrix(int(m)) for m in plan.matrix_indices], p.get_clv,
                  p.get_scaler, lnl, ps, p.get_sumtable(st), d, exact_lnl=False,
                  mfma_lnl=(S == 20 and aa_path != "vector-kernels" and R in (1, 2, 4)),
                  clv_exact=not (S == 20 and aa_path == "default"))
    if S == 20:
        assert p.scaling_certificate()["uncertified"] == 0
    p.destroy()
