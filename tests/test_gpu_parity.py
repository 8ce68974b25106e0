"""GPU parity tests proper: the HIP path, called through the C-ABI exactly like a
reference client, against the oracle on the same seeded inputs.

Bar (north_star): scaler counts bit-exact; lnL within 1e-10 relative.  What is
actually asserted is much tighter: P-matrices, every CLV, every scale buffer
and every per-site lnL bit-identical for 4- and 20-state data; the lnL sum
within 1e-12 relative (GPU adds sites in a tree, the reference sequentially).
"""
import numpy as np
import pytest

from helpers import (make_case, odd_state_case, many_state_case, build_partition, oracle_run, bits_equal, rel_err,
                     sumtable_err, invariant_of, clv_ok, clvs_bitwise)
from libpll_amd import workload as W
from libpll_amd.pllapi import (ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, SCALE_BUFFER_NONE,
                               PllError)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("dna_path")]

LNL_RTOL = 1e-12      # lnL (sum over sites), relative
PERSITE_RTOL = 1e-13  # per-site lnL, relative (in practice bit-identical)
DERIV_RTOL = 1e-10
# 20 states, default path: every CLV and scaler count is the reference's bit for bit (round 4: tip-inner ops too);
# the edge-lnL kernel on the matrix cores (likelihood_aa_mfma.hip) adds a row's 20 products as one fused chain
# where the reference uses four interleaved ones -> last-bit differences in the per-site lnL
MFMA_LNL_RTOL = 1e-11


def compare(p, o, case, R, exact=True):
    """exact=False: the 20-state matrix-core kernels -- scaler counts bit for bit, CLVs too (to 1e-13 below a tip-inner
    op of the whole-list kernel on the default path, round 6), lnL to MFMA_LNL_RTOL."""
    plan = case["plan"]
    clv_exact = clvs_bitwise(case["states"])
    for mi in plan.matrix_indices:
        assert bits_equal(p.get_pmatrix(int(mi)), o.pmat[int(mi)]), "P-matrix %d" % mi
    p.update_partials(plan.ops)
    o.update_partials()
    for op in plan.ops:
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        assert clv_ok(p.get_clv(node), o.clv[node], clv_exact), "CLV %d" % node
        if sc >= 0:
            assert (p.get_scaler(sc) == o.scalers[sc]).all(), "scaler %d" % sc
    if case["states"] == 20:
        assert p.scaling_certificate()["uncertified"] == 0
    lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
    lnl_o, ps_o = o.edge_loglikelihood(*plan.root_edge, persite=True)
    assert rel_err(ps, ps_o) < (PERSITE_RTOL if exact else MFMA_LNL_RTOL)
    assert abs(lnl - lnl_o) <= (LNL_RTOL if exact else MFMA_LNL_RTOL) * abs(lnl_o)
    return lnl


@pytest.mark.parametrize("states,shape,tips,sites", [
    (4, "balanced", 16, 1000), (4, "random", 23, 333), (4, "caterpillar", 60, 65),
    (20, "balanced", 8, 300), (20, "random", 11, 129)])
@pytest.mark.parametrize("pattern_tip", [0, ATTRIB_PATTERN_TIP])
@pytest.mark.parametrize("rate_scalers", [0, ATTRIB_RATE_SCALERS])
def test_evaluation_matches_oracle(gpu, orc, aa_mode, states, shape, tips, sites, pattern_tip,
                                   rate_scalers):
    if states == 4 and aa_mode != "exact":
        pytest.skip("mode only affects 20-state kernels")
    exact = states == 4 or aa_mode == "exact"
    attrs = pattern_tip | rate_scalers
    case = make_case(states, shape, tips, sites, seed=sites)
    if states == 20:
        case["rates"], case["freqs"] = gpu.aa_model("lg")
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    compare(p, o, case, 4, exact)
    # derivative pair at the root edge
    e = case["plan"].root_edge
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * 4, st)
    so = o.sumtable(e[0], e[2], e[1], e[3])
    assert sumtable_err(p.get_sumtable(st), so) < (1e-12 if exact else 1e-10)
    for t in (0.003, 0.13, 2.0):
        assert rel_err(p.compute_likelihood_derivatives(e[1], e[3], t, [0] * 4, st),
                       o.derivatives(so, t)) < DERIV_RTOL
    p.destroy()


@pytest.mark.parametrize("states,tips,expect_min", [(4, 700, 4), (20, 400, 5)])
@pytest.mark.parametrize("rate_scalers", [0, ATTRIB_RATE_SCALERS])
def test_deep_tree_scaler_counts_bit_exact(gpu, orc, aa_mode, states, tips, expect_min, rate_scalers):
    """Pattern-tip caterpillar: tip-inner kernels all the way down."""
    if states == 4 and aa_mode != "exact":
        pytest.skip("mode only affects 20-state kernels")
    exact = states == 4 or aa_mode == "exact"
    attrs = ATTRIB_PATTERN_TIP | rate_scalers
    case = make_case(states, "caterpillar", tips, 8, seed=5, alpha=0.5, branch=0.5, weights=False,
                     ambiguity=False, gap_frac=0.0)
    if states == 20:
        case["rates"], case["freqs"] = gpu.aa_model("lg")
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    compare(p, o, case, 4, exact)
    last = int(case["plan"].ops[-1]["parent_scaler_index"])
    assert p.get_scaler(last).min() >= expect_min
    p.destroy()


@pytest.mark.parametrize("rate_scalers", [0, ATTRIB_RATE_SCALERS])
def test_deep_aa_inner_inner_scalers(gpu, orc, aa_mode, rate_scalers):
    """Tip-CLV caterpillar: every op is inner-inner, so the 20-state matrix-core
    kernel carries the scaling all the way down; scaler counts must still be
    bit-exact in both modes."""
    case = make_case(20, "caterpillar", 300, 40, seed=8, alpha=0.5, branch=0.5, weights=False,
                     ambiguity=False, gap_frac=0.0)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    p = build_partition(gpu, case, rate_scalers)
    o = oracle_run(orc, gpu, p, case, rate_scalers)
    compare(p, o, case, 4, aa_mode == "exact")
    last = int(case["plan"].ops[-1]["parent_scaler_index"])
    assert p.get_scaler(last).min() >= 3
    p.destroy()


@pytest.mark.parametrize("rate_cats", [1, 2, 3, 8, 16])
@pytest.mark.parametrize("states", [4, 20])
def test_rate_category_counts(gpu, orc, aa_mode, states, rate_cats):
    """1/2/8/16 use the lane-per-(site,rate) kernels, 3 the generic fallback."""
    if states == 4 and aa_mode != "exact":
        pytest.skip("mode only affects 20-state kernels")
    exact = states == 4 or aa_mode == "exact"
    attrs = ATTRIB_PATTERN_TIP
    case = make_case(states, "random", 9, 101, rate_cats=rate_cats, seed=rate_cats)
    if states == 20:
        case["rates"], case["freqs"] = gpu.aa_model("wag")
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    compare(p, o, case, rate_cats, exact)
    p.destroy()


@pytest.mark.parametrize("shape,tips,sites,pattern_tip,expect_min", [
    ("caterpillar", 400, 8, ATTRIB_PATTERN_TIP, 5), ("caterpillar", 300, 40, 0, 3), ("random", 40, 333, ATTRIB_PATTERN_TIP, 0),
    ("balanced", 32, 1029, 0, 0)])
@pytest.mark.parametrize("rate_scalers", [0, ATTRIB_RATE_SCALERS])
@pytest.mark.parametrize("in_place", [False, True])
@pytest.mark.parametrize("rate_cats", [3, 7, 8, 12])
def test_chunked_rate_categories_20_states(gpu, orc, aa_mode, monkeypatch, shape, tips, sites, pattern_tip, expect_min,
                                           rate_scalers, in_place, rate_cats):
    """20 states with a number of rate categories other than 1, 2, 4: on the default path every op is SEVERAL launches
    of the matrix-core kernels, each over a chunk of 4, 2 or 1 categories (partials_aa_mfma.hip, SPLIT / CHUNK: 3 = 2 +
    1, 7 = 4 + 2 + 1, 8 = 4 + 4, 12 = 4 + 4 + 4), the per-site scaling verdict of the chunks so far travelling in the
    parent's scale buffer -- or in a scratch array when the op scales in place (parent scale buffer = the inner
    child's, in_place).  CLVs and scaler counts bit for bit, with scaling events, in both scaling modes, with waves that
    walk several tiles; the edge lnL (chunk by chunk too, likelihood_aa_mfma.hip) to the default path's tolerance."""
    if in_place and shape != "caterpillar":
        pytest.skip("one scale buffer all the way down needs a ladder")
    if rate_cats in (3, 12) and shape in ("random", "balanced") and rate_scalers:
        pytest.skip("covered by the other counts")
    import dataclasses
    monkeypatch.setenv("PLLHIP_AA_GRID_CAP", "3")
    case = make_case(20, shape, tips, sites, rate_cats=rate_cats, seed=tips, alpha=0.5,
                     branch=0.5 if shape == "caterpillar" else None, weights=False, ambiguity=shape != "caterpillar",
                     gap_frac=0.0 if shape == "caterpillar" else 0.05)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    if in_place:
        plan = case["plan"]
        ops = plan.ops.copy()
        for f in ("parent_scaler_index", "child1_scaler_index", "child2_scaler_index"):
            ops[f] = np.where(ops[f] >= 0, 0, ops[f])
        edge = tuple(0 if (i in (1, 3) and v >= 0) else v for i, v in enumerate(plan.root_edge))
        case["plan"] = dataclasses.replace(plan, ops=ops, root_edge=edge)
    attrs = pattern_tip | rate_scalers
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    compare(p, o, case, rate_cats, aa_mode == "exact")
    last = int(case["plan"].ops[-1]["parent_scaler_index"])
    assert p.get_scaler(last).min() >= expect_min
    p.destroy()


@pytest.mark.parametrize("rate_cats", [3, 6, 8])
@pytest.mark.parametrize("pattern_tip", [0, ATTRIB_PATTERN_TIP])
@pytest.mark.parametrize("rate_scalers", [0, ATTRIB_RATE_SCALERS])
def test_chunked_rate_categories_lnl_and_derivatives(gpu, orc, aa_mode, monkeypatch, rate_cats, pattern_tip, rate_scalers):
    """The calls behind the CLV updates on such partitions: edge lnL over an inner-inner and over a tip-inner edge,
    root lnL, with invariant sites and pattern weights; sumtable (chunk launches too) and derivatives."""
    monkeypatch.setenv("PLLHIP_AA_GRID_CAP", "2")
    exact = aa_mode == "exact"
    attrs = pattern_tip | rate_scalers
    case = make_case(20, "random", 12, 700, rate_cats=rate_cats, seed=rate_cats + 40, alpha=0.6)
    seqs = [bytearray(s) for s in case["seqs"]]
    for col in range(0, 700, 3):
        for s in seqs:
            s[col] = seqs[0][col]
    case["seqs"] = [bytes(s) for s in seqs]
    case["rates"], case["freqs"] = gpu.aa_model("wag")
    p = build_partition(gpu, case, attrs, pinv=0.25)
    o = oracle_run(orc, gpu, p, case, attrs, pinv=0.25)
    R = rate_cats
    compare(p, o, case, R, exact)
    plan = case["plan"]
    tol = PERSITE_RTOL if exact else MFMA_LNL_RTOL
    # an edge with a tip at one end: the last tip-inner (or, with tips as CLVs, any) op's parent and its tip child
    op = [q for q in plan.ops if int(q["child2_clv_index"]) < plan.tips or int(q["child1_clv_index"]) < plan.tips][-1]
    tipside = 1 if int(op["child1_clv_index"]) < plan.tips else 2
    edge = (int(op["parent_clv_index"]), int(op["parent_scaler_index"]), int(op["child%d_clv_index" % tipside]), -1,
            int(op["child%d_matrix_index" % tipside]))
    lnl, ps = p.compute_edge_loglikelihood(*edge, [0] * R, persite=True)
    lnl_o, ps_o = o.edge_loglikelihood(*edge, persite=True)
    assert rel_err(ps, ps_o) < tol and abs(lnl - lnl_o) <= tol * abs(lnl_o)
    e = plan.root_edge
    if not rate_scalers:
        # root lnL (core_likelihood.c:25-209) restated on the CLV the device holds
        lnl, ps = p.compute_root_loglikelihood(e[0], e[1], [0] * R, persite=True)
        clv, sc, inv = p.get_clv(e[0]), p.get_scaler(e[1]).astype(np.float64), invariant_of(p)
        fr = np.asarray(case["freqs"], dtype=np.float64)
        inv_lk = np.where(inv >= 0, fr[np.maximum(inv, 0)], 0.0) * 0.25
        site = ((clv @ fr) * 0.75 + inv_lk[:, None]).sum(axis=1) / R
        ps_o = (np.log(site) + sc * np.log(2.0 ** -256)) * case["pw"]
        assert rel_err(ps, ps_o) < 1e-11 and abs(lnl - ps_o.sum()) <= 1e-11 * abs(lnl)
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    so = o.sumtable(e[0], e[2], e[1], e[3])
    assert sumtable_err(p.get_sumtable(st), so) < (1e-12 if exact else 1e-10)
    for t in (0.003, 0.2):
        assert rel_err(p.compute_likelihood_derivatives(e[1], e[3], t, [0] * R, st), o.derivatives(so, t)) < DERIV_RTOL
    p.destroy()


@pytest.mark.parametrize("states", [5, 7])
@pytest.mark.parametrize("pattern_tip", [0, ATTRIB_PATTERN_TIP])
def test_odd_state_counts(gpu, orc, states, pattern_tip):
    case = odd_state_case(states)
    p = build_partition(gpu, case, pattern_tip)
    o = oracle_run(orc, gpu, p, case, pattern_tip)
    compare(p, o, case, 4)
    p.destroy()


# state counts without a dedicated kernel (partials_gen_tile.hip): rows-in-registers kernels
# (up to 16 states with a power-of-two rate_cats), P-rows-in-registers kernels (9..64 states,
# 16 / 32 / 64 lanes per site) and the LDS-tiled kernels (up to 8 states with any other rate_cats)
GENERIC_SHAPES = [(2, 4), (3, 4), (5, 4), (6, 2), (7, 1), (8, 8), (5, 3), (9, 4), (13, 4), (11, 16), (16, 2),
                  (13, 3), (17, 3), (21, 4), (32, 4), (40, 1), (50, 2), (61, 4), (64, 2),
                  (7, 64), (12, 40)]  # P tables too large for the rows kernel: LDS-tiled / P-row kernels


@pytest.mark.parametrize("states,rate_cats", GENERIC_SHAPES)
@pytest.mark.parametrize("pattern_tip", [0, ATTRIB_PATTERN_TIP])
@pytest.mark.parametrize("rate_scalers", [0, ATTRIB_RATE_SCALERS])
def test_generic_state_kernels(gpu, orc, states, rate_cats, pattern_tip, rate_scalers):
    """Several tiles / rounds per op, all three op kinds, both scaling modes; bit-exact
    against the plain-C order of the reference."""
    attrs = pattern_tip | rate_scalers
    if states > 32:
        if pattern_tip:
            pytest.skip("tip characters are 32-bit state masks")
        case = many_state_case(states, tips=7, sites=75, seed=states, rate_cats=rate_cats)
    else:
        case = odd_state_case(states, tips=10, sites=333, seed=states + rate_cats, rate_cats=rate_cats)
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    compare(p, o, case, rate_cats)
    e = case["plan"].root_edge
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * rate_cats, st)
    so = o.sumtable(e[0], e[2], e[1], e[3])
    assert sumtable_err(p.get_sumtable(st), so) < 1e-12
    assert rel_err(p.compute_likelihood_derivatives(e[1], e[3], 0.13, [0] * rate_cats, st),
                   o.derivatives(so, 0.13)) < DERIV_RTOL
    p.destroy()


@pytest.mark.parametrize("states,rate_cats", [(5, 4), (5, 3), (13, 3), (24, 2), (61, 1), (7, 64)])
def test_generic_state_ragged_site_counts(gpu, orc, states, rate_cats):
    """Site counts around the group sizes of the kernels (a wave's 64 / rate_cats sites, 16..64-site groups)."""
    for sites in (1, 2, 15, 17, 47, 48, 49, 63, 64, 65, 130):
        for attrs in ((0, ATTRIB_RATE_SCALERS) if states > 32 else (ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS)):
            kw = dict(tips=6, sites=sites, seed=sites + states, rate_cats=rate_cats)
            case = many_state_case(states, **kw) if states > 32 else odd_state_case(states, **kw)
            p = build_partition(gpu, case, attrs)
            o = oracle_run(orc, gpu, p, case, attrs)
            compare(p, o, case, rate_cats)
            p.destroy()


@pytest.mark.parametrize("states,rate_cats,tips", [(2, 4, 900), (5, 4, 500), (5, 3, 500), (13, 4, 400), (13, 3, 400),
                                                   (24, 2, 350), (61, 4, 300)])
@pytest.mark.parametrize("rate_scalers", [0, ATTRIB_RATE_SCALERS])
def test_generic_state_deep_scaling(gpu, orc, states, rate_cats, tips, rate_scalers):
    """Caterpillars deep enough for several scaling events per site, tips as CLVs (all
    ops inner-inner) and as characters (tip-inner all the way down)."""
    for pattern_tip in ((0,) if states > 32 else (0, ATTRIB_PATTERN_TIP)):
        attrs = pattern_tip | rate_scalers
        kw = dict(tips=tips, sites=9, seed=5, shape="caterpillar", rate_cats=rate_cats, alpha=0.5,
                  branch=0.5, weights=False, gap_frac=0.0)
        case = many_state_case(states, **kw) if states > 32 else odd_state_case(states, **kw)
        p = build_partition(gpu, case, attrs)
        o = oracle_run(orc, gpu, p, case, attrs)
        compare(p, o, case, rate_cats)
        last = int(case["plan"].ops[-1]["parent_scaler_index"])
        assert p.get_scaler(last).min() >= 2
        p.destroy()


@pytest.mark.parametrize("sites", [1, 2, 15, 16, 17, 63, 64, 65, 255, 257])
def test_ragged_site_counts(gpu, orc, sites):
    """Site counts around the wave (64) and quad (16 sites x 4 rates) boundaries."""
    for attrs in (0, ATTRIB_PATTERN_TIP | ATTRIB_RATE_SCALERS):
        case = make_case(4, "random", 6, sites, seed=sites, ambiguity=sites > 30)
        p = build_partition(gpu, case, attrs)
        o = oracle_run(orc, gpu, p, case, attrs)
        compare(p, o, case, 4)
        p.destroy()


@pytest.mark.parametrize("states", [4, 20])
def test_invariant_sites_model(gpu, orc, aa_mode, states):
    if states == 4 and aa_mode != "exact":
        pytest.skip("mode only affects 20-state kernels")
    exact = states == 4 or aa_mode == "exact"
    attrs = ATTRIB_PATTERN_TIP
    case = make_case(states, "random", 7, 120, seed=9, gap_frac=0.0, ambiguity=False)
    seqs = [bytearray(s) for s in case["seqs"]]
    for col in range(0, 120, 3):
        for s in seqs:
            s[col] = seqs[0][col]
    case["seqs"] = [bytes(s) for s in seqs]
    if states == 20:
        case["rates"], case["freqs"] = gpu.aa_model("wag")
    p = build_partition(gpu, case, attrs, pinv=0.3)
    o = oracle_run(orc, gpu, p, case, attrs, pinv=0.3)
    compare(p, o, case, 4, exact)
    e = case["plan"].root_edge
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * 4, st)
    so = o.sumtable(e[0], e[2], e[1], e[3])
    assert rel_err(p.compute_likelihood_derivatives(e[1], e[3], 0.2, [0] * 4, st),
                   o.derivatives(so, 0.2)) < DERIV_RTOL
    p.destroy()


def test_zero_and_tiny_branch_lengths(gpu, orc):
    """t = 0 must give the exact identity; tiny t exercises the expm1 trick
    (the reference's pmatrix test, test/src/pmatrix.c)."""
    case = make_case(4, "balanced", 8, 64, seed=2)
    case["plan"].branch_lengths[:6] = [0.0, 1e-12, 1e-9, 1e-6, 50.0, 900.0]
    p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    o = oracle_run(orc, gpu, p, case, ATTRIB_PATTERN_TIP)
    m0 = p.get_pmatrix(int(case["plan"].matrix_indices[0]))
    assert bits_equal(m0, np.broadcast_to(np.eye(4), (4, 4, 4)))
    compare(p, o, case, 4)
    p.destroy()


def test_no_scale_buffers(gpu, orc):
    """PLL_SCALE_BUFFER_NONE everywhere, as all self-contained reference tests use."""
    case = make_case(4, "random", 12, 200, seed=4)
    ops = case["plan"].ops
    for f in ("parent_scaler_index", "child1_scaler_index", "child2_scaler_index"):
        ops[f] = SCALE_BUFFER_NONE
    e = case["plan"].root_edge
    case["plan"].root_edge = (e[0], SCALE_BUFFER_NONE, e[2], SCALE_BUFFER_NONE, e[4])
    p = build_partition(gpu, case, 0)
    o = oracle_run(orc, gpu, p, case, 0)
    compare(p, o, case, 4)
    p.destroy()


def test_tip_inner_equals_inner_inner(gpu):
    """The same tree evaluated with pattern tips (tip-tip / tip-inner CLV kernels,
    tip-inner lnL kernel at the caterpillar's root edge) and with tip CLVs (only
    inner-inner kernels) gives the identical lnL -- the ii-vs-ti equality
    test/src/scaling.c checks on the reference."""
    for shape in ("balanced", "caterpillar"):
        case = make_case(4, shape, 8, 500, seed=6, weights=False)
        plan = case["plan"]
        p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
        p.update_partials(plan.ops)
        lnl_pt, ps_pt = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
        q = build_partition(gpu, case, 0)
        q.update_partials(plan.ops)
        lnl_clv, ps_clv = q.compute_edge_loglikelihood(*plan.root_edge, [0] * 4, persite=True)
        assert bits_equal(ps_pt, ps_clv)
        assert lnl_pt == lnl_clv
        p.destroy()
        q.destroy()


def test_rccl_allreduce_path_single_rank(gpu):
    """The multi-GPU code path on one GPU: a 1-rank RCCL communicator is created
    through pll_amd_comm_unique_id / pll_amd_comm_init and every lnL / derivative
    result then goes through ncclAllReduce on the partition's stream.  With one
    rank the sum must leave the values unchanged."""
    import ctypes
    case = make_case(4, "random", 9, 3000, seed=12)
    plan = case["plan"]
    p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    p.update_partials(plan.ops)
    lnl0 = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
    e = plan.root_edge
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * 4, st)
    d0 = p.compute_likelihood_derivatives(e[1], e[3], 0.2, [0] * 4, st)
    uid = ctypes.create_string_buffer(128)
    assert gpu.lib.pll_amd_comm_unique_id(uid), gpu.errmsg()
    p.comm_init(0, 1, uid.raw)
    assert p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4) == lnl0
    assert p.compute_likelihood_derivatives(e[1], e[3], 0.2, [0] * 4, st) == d0
    p.destroy()


def test_error_paths(gpu):
    """Same error behaviour as the reference where it defines one."""
    with pytest.raises(PllError):
        gpu.partition_create(4, 2, 4, 10, 1, 5, 4, 2, 1 | 2)   # two ISA flags (pll.c:413-418)
    assert gpu.errno() == 113
    p = gpu.partition_create(4, 2, 4, 6, 1, 5, 4, 2, ATTRIB_PATTERN_TIP)
    with pytest.raises(PllError):
        p.set_tip_states(0, gpu.map("nt"), b"ACGT!A")           # illegal character (pll.c:836-841)
    assert gpu.errno() == 114
    with pytest.raises(PllError):
        p.set_tip_clv(0, np.zeros(24))                            # pll.c:1008-1014
    assert gpu.errno() == 115
    with pytest.raises(PllError):
        p.update_invariant_sites_proportion(0, 1.5)
    assert gpu.errno() == 118
    p.destroy()


# ---------------------------------------------------------------- full size

@pytest.mark.parametrize("scale_attr", [0, ATTRIB_RATE_SCALERS], ids=["per-site-scalers", "per-rate-scalers"])
def test_full_size_properties(gpu, scale_attr):
    """(per-rate scale buffers -- PLL_ATTRIB_RATE_SCALERS, the reference's per-rate rule core_partials_avx.c:494-503 --
    reach the whole-list kernel at full size here; until round 5 only at <= 1,029 sites)
    BASELINE.json configs[1] (1,000,000 sites x 4 rates, 64 taxa): properties
    that hold at any size, checked on the real workload:
      * sum(per-site lnL) == lnL                    (reduction is complete)
      * lnL(all) == lnL(first half) + lnL(second half)   (sites independent; this
        is also the site-sharding identity the multi-GPU path relies on)
      * doubling every pattern weight doubles lnL   (linearity)
      * tip-CLV mode (all inner-inner) == pattern-tip mode, per site, bitwise
      * re-running the same traversal is bitwise reproducible
    """
    sites, T, R = 1_000_000, 64, 4
    plan = W.balanced_tree(T, seed=42)
    seqs = W.simulated_alignment(plan, sites, W.GTR_RATES, W.GTR_FREQS,
                                 gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    fi = [0] * R
    p = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP | scale_attr)
    p.update_partials(plan.ops)
    lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
    assert np.isfinite(lnl) and lnl < 0
    assert abs(ps.sum() - lnl) <= 1e-11 * abs(lnl)
    p.update_partials(plan.ops)
    lnl2, ps2 = p.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
    assert lnl2 == lnl and bits_equal(ps, ps2)
    p.set_pattern_weights(np.full(sites, 2, dtype=np.uint32))
    lnl_w2 = p.compute_edge_loglikelihood(*plan.root_edge, fi)
    assert abs(lnl_w2 - 2 * lnl) <= 1e-12 * abs(lnl)
    p.destroy()

    halves = []
    for lo, hi in ((0, sites // 2), (sites // 2, sites)):
        h = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP, site_range=(lo, hi))
        h.update_partials(plan.ops)
        v, hps = h.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
        assert bits_equal(hps, ps[lo:hi])
        halves.append(v)
        h.destroy()
    assert abs(sum(halves) - lnl) <= 1e-12 * abs(lnl)

    n = 250_000   # tip-CLV mode needs 2x the CLV memory traffic; a quarter is plenty
    q = W.setup_partition(gpu, plan, seqs, 4, R, scale_attr, site_range=(0, n))
    q.update_partials(plan.ops)
    _, qps = q.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
    assert bits_equal(qps, ps[:n])
    q.destroy()


def test_full_size_properties_20_states(gpu, monkeypatch):
    """BASELINE.json configs[2] (20 states, 4 rates, 200,000 sites, 64 taxa) on the
    matrix-core kernels: reduction complete, halves add up to the whole (per-site
    values of a half bitwise equal to the whole's -- a tile never mixes sites), run to
    run reproducible, scaler counts identical to the bit-exact vector kernels' and lnL
    within the stated tolerance of theirs."""
    sites, T, R = 200_000, 64, 4
    plan = W.balanced_tree(T, seed=42)
    rates, freqs = gpu.aa_model("lg")
    seqs = W.simulated_alignment(plan, sites, rates, freqs, gpu.compute_gamma_cats(W.GAMMA_ALPHA, R),
                                 seed=42)
    fi = [0] * R
    monkeypatch.setenv("PLLHIP_AA_EXACT", "0")
    p = W.setup_partition(gpu, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
    p.update_partials(plan.ops)
    lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
    assert np.isfinite(lnl) and lnl < 0
    assert abs(ps.sum() - lnl) <= 1e-11 * abs(lnl)
    p.update_partials(plan.ops)
    lnl2, ps2 = p.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
    assert lnl2 == lnl and bits_equal(ps, ps2)
    scalers = {int(op["parent_scaler_index"]): p.get_scaler(int(op["parent_scaler_index"]))
               for op in (plan.ops[-1], plan.ops[-2], plan.ops[-4], plan.ops[-8], plan.ops[-16])}

    halves = []
    for lo, hi in ((0, 99_984), (99_984, sites)):     # a multiple of 16 sites: whole tiles
        h = W.setup_partition(gpu, plan, seqs, 20, R, ATTRIB_PATTERN_TIP, site_range=(lo, hi))
        h.update_partials(plan.ops)
        v, hps = h.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
        assert bits_equal(hps, ps[lo:hi])
        halves.append(v)
        h.destroy()
    assert abs(sum(halves) - lnl) <= 1e-12 * abs(lnl)
    p.destroy()

    # the bit-exact kernels on the WHOLE alignment: per-site lnL within the stated tolerance, and
    # the scaler counts of every level's last op identical (a product within an ulp of 2^-256
    # could in principle scale on one side only: it does not happen on these 200,000 sites)
    monkeypatch.setenv("PLLHIP_AA_EXACT", "1")
    e = W.setup_partition(gpu, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
    e.update_partials(plan.ops)
    _, eps = e.compute_edge_loglikelihood(*plan.root_edge, fi, persite=True)
    assert rel_err(ps, eps) < MFMA_LNL_RTOL
    for op in (plan.ops[-1], plan.ops[-2], plan.ops[-4], plan.ops[-8], plan.ops[-16]):
        sc = int(op["parent_scaler_index"])
        assert (e.get_scaler(sc) == scalers[sc]).all(), "scaler %d" % sc
    e.destroy()


def test_full_size_against_reference_sample(gpu, ref):
    """A 100,000-site slice of the full workload through the genuine reference
    (AVX2 flag) and through the HIP path: scalers and per-site lnL bitwise."""
    from libpll_amd.pllapi import ATTRIB_ARCH_AVX2
    sites, T, R = 100_000, 64, 4
    plan = W.balanced_tree(T, seed=42)
    seqs = W.simulated_alignment(plan, sites, W.GTR_RATES, W.GTR_FREQS,
                                 gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    a = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP)
    r = W.setup_partition(ref, plan, seqs, 4, R, ATTRIB_PATTERN_TIP | ATTRIB_ARCH_AVX2)
    a.update_partials(plan.ops)
    r.update_partials(plan.ops)
    for op in plan.ops[-3:]:
        assert (a.get_scaler(int(op["parent_scaler_index"])) ==
                r.get_scaler(int(op["parent_scaler_index"]))).all()
        assert bits_equal(a.get_clv(int(op["parent_clv_index"])), r.get_clv(int(op["parent_clv_index"])))
    la, pa = a.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
    lr, pr = r.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
    assert rel_err(pa, pr) < PERSITE_RTOL
    assert abs(la - lr) <= 1e-10 * abs(lr)
    a.destroy()
    r.destroy()
