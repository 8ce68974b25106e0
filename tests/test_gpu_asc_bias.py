"""Ascertainment-bias correction (likelihood.c:24-119,170-247,
321-414, core_derivatives.c:654-727) on the GPU against the genuine reference
library: a partition with `states` extra per-state sites, the three correction
types, edge (inner-inner and tip-inner) and root log-likelihood, sumtable and
derivatives, scalers that fire on a deep tree, state weights.  The reference's own
test for this (test/src/asc-bias.c) needs data files that are not in the snapshot."""
import numpy as np
import pytest

from helpers import make_case, bits_equal, rel_err, sumtable_err, case_map
from libpll_amd.pllapi import (ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, ATTRIB_ARCH_AVX2,
                               ATTRIB_AB_LEWIS, ATTRIB_AB_FELSENSTEIN, ATTRIB_AB_STAMATAKIS,
                               ATTRIB_AB_FLAG, PllError)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("dna_path")]

TYPES = {"lewis": ATTRIB_AB_LEWIS, "felsenstein": ATTRIB_AB_FELSENSTEIN,
         "stamatakis": ATTRIB_AB_STAMATAKIS}


def build(lib, case, attrs, state_weights=None):
    plan, S, R = case["plan"], case["states"], case["rate_cats"]
    if lib.is_amd:
        attrs &= ~0xF
    p = lib.partition_create(plan.tips, plan.clv_buffers, S, case["sites"], 1, plan.prob_matrices,
                             R, plan.scale_buffers, attrs)
    p.set_frequencies(0, case["freqs"])
    p.set_subst_params(0, case["rates"])
    p.set_category_rates(lib.compute_gamma_cats(case["alpha"], R))
    cmap = case_map(lib, case)
    for i, s in enumerate(case["seqs"]):
        p.set_tip_states(i, cmap, s)
    p.set_pattern_weights(case["pw"])
    if state_weights is not None:
        p.set_asc_state_weights(state_weights)
    p.update_prob_matrices([0] * R, plan.matrix_indices, plan.branch_lengths)
    return p


def both(gpu, ref, case, attrs, sw):
    # the product follows the reference's AVX2-flag operation order for 4 and 20
    # states and the plain-C order for every other state count
    arch = ATTRIB_ARCH_AVX2 if case["states"] in (4, 20) else 0
    return build(gpu, case, attrs, sw), build(ref, case, attrs | arch, sw)


@pytest.mark.parametrize("kind", sorted(TYPES))
@pytest.mark.parametrize("states,pattern_tip,rate_scalers",
                         [(4, True, False), (4, False, False), (4, True, True), (20, False, False),
                          (5, False, False)])
def test_asc_bias_matches_reference(gpu, ref, monkeypatch, kind, states, pattern_tip, rate_scalers):
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)   # (the default path: 20 states on the matrix cores)
    from helpers import odd_state_case
    if states in (4, 20):
        case = make_case(states, "random", 12, 157, seed=states + len(kind))
    else:
        case = odd_state_case(states, tips=9, sites=41, seed=5)
        case["pw"] = np.ones(41, dtype=np.uint32)
    plan, R = case["plan"], case["rate_cats"]
    attrs = TYPES[kind] | (ATTRIB_PATTERN_TIP if pattern_tip else 0) | \
        (ATTRIB_RATE_SCALERS if rate_scalers else 0)
    rng = np.random.default_rng(3)
    sw = rng.integers(1, 40, size=states).astype(np.uint32) if kind != "lewis" else None
    g, r = both(gpu, ref, case, attrs, sw)
    assert g.s.asc_bias_alloc == 1 and g.sites_total == case["sites"] + states
    g.update_partials(plan.ops)
    r.update_partials(plan.ops)
    for op in plan.ops:
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        assert bits_equal(g.get_clv(node), r.get_clv(node)), "CLV %d (with its extra sites)" % node
        assert (g.get_scaler(sc) == r.get_scaler(sc)).all()
    e = plan.root_edge
    fi = [0] * R
    lg, psg = g.compute_edge_loglikelihood(*e, fi, persite=True)
    lr, psr = r.compute_edge_loglikelihood(*e, fi, persite=True)
    assert abs(lg - lr) <= 1e-11 * abs(lr), (lg, lr)
    assert rel_err(psg, psr) < 1e-12
    # the correction really is part of the value
    g.set_asc_bias_type(0)
    plain = g.compute_edge_loglikelihood(*e, fi)
    g.set_asc_bias_type(TYPES[kind])
    assert abs(plain - lg) > 1e-6 * abs(lg)
    # a tip-inner edge and the root form
    if pattern_tip:
        op = next(o for o in plan.ops if int(o["child1_clv_index"]) < plan.tips)
        te = (int(op["parent_clv_index"]), int(op["parent_scaler_index"]), int(op["child1_clv_index"]),
              -1, int(op["child1_matrix_index"]))
        # (not a likelihood of the whole tree, just the same function of the same buffers)
        a, b = g.compute_edge_loglikelihood(*te, fi), r.compute_edge_loglikelihood(*te, fi)
        assert abs(a - b) <= 1e-11 * abs(b), (a, b)
    a = g.compute_root_loglikelihood(e[0], e[1], fi)
    b = r.compute_root_loglikelihood(e[0], e[1], fi)
    assert abs(a - b) <= 1e-11 * abs(b), (a, b)
    # sumtable (over sites + states rows) and derivatives
    stg, strf = g.alloc_sumtable(), r.alloc_sumtable()
    g.update_sumtable(e[0], e[2], e[1], e[3], fi, stg)
    r.update_sumtable(e[0], e[2], e[1], e[3], fi, strf)
    assert sumtable_err(g.get_sumtable(stg), r.get_sumtable(strf)) < 1e-11
    for t in (0.03, 0.4, 2.0):
        dg = g.compute_likelihood_derivatives(e[1], e[3], t, fi, stg)
        dr = r.compute_likelihood_derivatives(e[1], e[3], t, fi, strf)
        assert rel_err(np.array(dg), np.array(dr)) < 1e-9, (t, dg, dr)
    g.destroy()
    r.destroy()


@pytest.mark.parametrize("kind", ["lewis", "stamatakis"])
def test_asc_bias_with_scaling_events(gpu, ref, kind):
    """A 500-tip ladder with long branches: the extra sites' scaler counts are 2-3 and
    enter the correction through pow(2^-256, count) / count*log(2^-256)."""
    case = make_case(4, "caterpillar", 500, 24, seed=8, branch=10.0, ambiguity=False, gap_frac=0.0)
    plan, R = case["plan"], 4
    sw = np.arange(1, 5, dtype=np.uint32)
    g, r = both(gpu, ref, case, TYPES[kind] | ATTRIB_PATTERN_TIP, sw)
    g.update_partials(plan.ops)
    r.update_partials(plan.ops)
    last = int(plan.ops[-1]["parent_scaler_index"])
    assert r.get_scaler(last)[-4:].min() >= 1, "extra sites no longer scale"
    assert (g.get_scaler(last) == r.get_scaler(last)).all()
    e = plan.root_edge
    a, b = g.compute_edge_loglikelihood(*e, [0] * R), r.compute_edge_loglikelihood(*e, [0] * R)
    assert abs(a - b) <= 1e-11 * abs(b), (a, b)
    stg, strf = g.alloc_sumtable(), r.alloc_sumtable()
    g.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, stg)
    r.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, strf)
    dg = g.compute_likelihood_derivatives(e[1], e[3], 0.6, [0] * R, stg)
    dr = r.compute_likelihood_derivatives(e[1], e[3], 0.6, [0] * R, strf)
    assert rel_err(np.array(dg), np.array(dr)) < 1e-9, (dg, dr)
    g.destroy()
    r.destroy()


def test_asc_bias_api_contract(gpu, ref):
    """Error behaviour of pll.c:1061-1116 and models.c:407-414."""
    case = make_case(4, "balanced", 8, 50, seed=2)
    plain = build(gpu, case, ATTRIB_PATTERN_TIP)
    with pytest.raises(PllError) as ei:
        plain.set_asc_bias_type(ATTRIB_AB_LEWIS)
    assert "not created with ascertainment bias support" in str(ei.value)
    plain.destroy()
    for lib in (gpu, ref):
        p = build(lib, case, ATTRIB_PATTERN_TIP | ATTRIB_AB_FLAG)
        assert p.s.asc_bias_alloc == 1 and (p.s.attributes & (7 << 5)) == 0
        with pytest.raises(PllError) as ei:
            p.set_asc_bias_type(5)
        assert "Illegal ascertainment bias algorithm" in str(ei.value) and lib.errno() == 121
        p.set_asc_bias_type(ATTRIB_AB_FELSENSTEIN)
        with pytest.raises(PllError) as ei:
            p.update_invariant_sites_proportion(0, 0.2)
        assert lib.errno() == 117
        p.destroy()
    # 20 states + pattern tips: refused here (the reference's extra tip characters are unusable)
    aa = make_case(20, "balanced", 8, 30, seed=3)
    with pytest.raises(PllError) as ei:
        build(gpu, aa, ATTRIB_PATTERN_TIP | ATTRIB_AB_LEWIS)
    assert "tip CLVs" in str(ei.value)


@pytest.mark.parametrize("kind", sorted(TYPES))
def test_asc_bias_is_additive_over_site_shards(gpu, kind):
    """What the multi-GPU path relies on: with the per-state extra sites
    present on every shard, the corrected lnL of the shards adds up to the corrected lnL of
    the whole alignment (state weights split the same way as the sites)."""
    case = make_case(4, "random", 10, 512, seed=21)
    plan, R = case["plan"], case["rate_cats"]
    sw = np.array([8, 2, 6, 4], dtype=np.uint32)
    whole = build(gpu, case, TYPES[kind] | ATTRIB_PATTERN_TIP, sw)
    whole.update_partials(plan.ops)
    total = whole.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
    whole.destroy()
    parts = 0.0
    for lo, hi in ((0, 256), (256, 512)):
        half = dict(case, sites=hi - lo, seqs=[s[lo:hi] for s in case["seqs"]], pw=case["pw"][lo:hi])
        p = build(gpu, half, TYPES[kind] | ATTRIB_PATTERN_TIP, sw // 2)
        p.update_partials(plan.ops)
        parts += p.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
        p.destroy()
    assert abs(parts - total) <= 1e-11 * abs(total), (kind, parts, total)
