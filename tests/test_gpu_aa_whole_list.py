"""20-state data: the whole-list kernel (partials_aa_fused.hip) and the reference-order matrix-core
products (round 3), through the C-ABI against the oracle and against the per-level launches.

What is asserted:
  * with PLLHIP_AA_TI_MFMA=0 EVERY CLV update equals the oracle bit for bit: inner-inner ops on the matrix cores --
    core_partials_avx2.c:632-750's four FMA chains and pairwise tree, reproduced with the MFMA's own
    accumulation order (tools/mfma_order_probe.hip) --, tip-inner ops (round 4) with their one mat-vec on the
    vector unit in the non-fused order of core_partials_avx.c:1229-1284, tip-tip ops and the table lookups
    that replace ops over tip-tip results; scaler counts equal everywhere;
  * on the DEFAULT path (round 6: the whole-list kernel runs the mat-vec of tip-inner ops on the matrix cores, behind
    the scaling certificate of tests/test_gpu_cert.py) every scaler count still equals the oracle's bit for bit, every
    CLV to 1e-13 entry by entry -- and bit for bit wherever no tip-inner op lies below it;
  * the whole-list kernel (PLLHIP_FUSED=2) and the per-level launches (PLLHIP_FUSED=0) give the same
    bits for every CLV and scale buffer: full traversals, partial traversals on top of them (operands
    from earlier calls), tips as characters and as CLVs, trees that need evictions, ragged site
    counts, lists without scale buffers;
  * waves that walk many tiles (PLLHIP_AA_GRID_CAP) on trees deep enough to scale: the regression
    test of a round-2 bug (from a wave's third tile on, inherited scaler counts were read at the
    second tile's sites) that no test saw because no test had sites x depth x scaling together.
"""
import numpy as np
import pytest

from helpers import make_case, build_partition, oracle_run, bits_equal, clv_ok
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, SCALE_BUFFER_NONE

pytestmark = pytest.mark.gpu


def _case(gpu, shape, tips, sites, seed=11):
    case = make_case(20, shape, tips, sites, seed=seed)
    case["rates"], case["freqs"] = gpu.aa_model("lg")
    return case


def _kinds(plan, tips, attrs):
    out = []
    for op in plan.ops:
        t1 = bool(attrs & ATTRIB_PATTERN_TIP) and int(op["child1_clv_index"]) < tips
        t2 = bool(attrs & ATTRIB_PATTERN_TIP) and int(op["child2_clv_index"]) < tips
        out.append("tt" if (t1 and t2) else "ti" if (t1 or t2) else "ii")
    return out


def _evaluate(gpu, case, attrs, monkeypatch, fused, partial=True, ti_mfma="0"):
    """ti_mfma: "0" = tip-inner mat-vecs of the whole-list kernel in the reference's order (every CLV bit for bit),
    "1" = on the matrix cores, the default."""
    monkeypatch.setenv("PLLHIP_AA_TI_MFMA", ti_mfma)
    monkeypatch.setenv("PLLHIP_AA_EXACT", "0")
    monkeypatch.setenv("PLLHIP_AA_CHERRY", "2")
    monkeypatch.setenv("PLLHIP_FUSED", fused)
    plan = case["plan"]
    p = build_partition(gpu, case, attrs)
    p.update_partials(plan.ops)
    if partial:
        # partial traversals on top: the last few ops again, their operands written by the call before
        p.update_partials(plan.ops[-3:])
        p.update_partials(plan.ops[-1:])
    clvs = [p.get_clv(int(op["parent_clv_index"])) for op in plan.ops]
    scs = [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops]
    lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * 4)
    return p, clvs, scs, lnl


@pytest.mark.parametrize("shape,tips,sites", [
    ("balanced", 16, 333), ("balanced", 64, 1000), ("random", 12, 500), ("caterpillar", 40, 100),
    ("random", 50, 97), ("balanced", 128, 64), ("random", 30, 1), ("random", 200, 40), ("balanced", 8, 40000)])
@pytest.mark.parametrize("attrs", [ATTRIB_PATTERN_TIP, 0])
@pytest.mark.parametrize("tt", [None, "1", "0"])
def test_whole_list_equals_per_level_and_oracle(gpu, orc, monkeypatch, shape, tips, sites, attrs, tt):
    """tt: where the list's tip-tip ops run -- None: ahead of the list when it is new, inside it when it comes again
    (the partial traversals of _evaluate make the full list come once only); "1": always inside, as lookups over the
    two tip tables; "0": always ahead."""
    if tt is not None:
        if attrs == 0:
            pytest.skip("no tip-tip ops without pattern tips")
        monkeypatch.setenv("PLLHIP_AA_TT_INSIDE", tt)
    case = _case(gpu, shape, tips, sites)
    plan = case["plan"]
    pf, cf, sf, lf = _evaluate(gpu, case, attrs, monkeypatch, "2")
    pf.destroy()
    pd, cd, sd, ld = _evaluate(gpu, case, attrs, monkeypatch, "2", ti_mfma="1")   # the default
    cert = pd.scaling_certificate()
    pd.destroy()
    pl, cl, sl, ll = _evaluate(gpu, case, attrs, monkeypatch, "0")
    o = oracle_run(orc, gpu, pl, case, attrs)
    o.update_partials()
    pl.destroy()
    kinds = _kinds(plan, tips, attrs)
    for op, kind, a, b, x, y, ad, xd in zip(plan.ops, kinds, cf, cl, sf, sl, cd, sd):
        node = int(op["parent_clv_index"])
        assert bits_equal(a, b), "CLV %d: whole list != per level" % node
        assert (x == y).all(), "scale buffer of CLV %d: whole list != per level" % node
        assert (x == o.scalers[int(op["parent_scaler_index"])]).all(), "scaler counts of CLV %d != oracle" % node
        # round 4: tip-inner ops too (their one mat-vec runs on the vector unit in the reference's non-fused order)
        assert bits_equal(a, o.clv[node]), "CLV %d (%s) != oracle" % (node, kind)
        # round 6, the default: counts bit for bit, CLVs to rounding
        assert (xd == x).all(), "default path: scaler counts of CLV %d != oracle" % node
        assert clv_ok(ad, o.clv[node], exact=False), "default path: CLV %d (%s)" % (node, kind)
    assert lf == ll
    ref = o.edge_loglikelihood(*plan.root_edge)
    assert abs(lf - ref) <= 1e-11 * abs(ref) and abs(ld - ref) <= 1e-11 * abs(ref)
    assert cert["uncertified"] == 0
    if "ti" in kinds and attrs:
        assert cert["lists"] >= 1 or all(int(op["parent_scaler_index"]) < 0 for op in plan.ops), cert


@pytest.mark.parametrize("tips,sites", [(24, 900), (40, 130)])
def test_whole_list_with_another_character_map(gpu, orc, monkeypatch, tips, sites):
    """20 states over a hand-made alphabet (A..T and a gap: 21 tip codes, not the protein map's 23):
    the pair and lookup tables of the list -- made by its prepare launch, one workgroup per (table, first character) --
    are indexed by the partition's OWN number of codes."""
    from helpers import odd_state_case
    case = odd_state_case(20, tips=tips, sites=sites, seed=tips)
    plan = case["plan"]
    attrs = ATTRIB_PATTERN_TIP
    pf, cf, sf, lf = _evaluate(gpu, case, attrs, monkeypatch, "2")
    assert pf.s.maxstates != 23
    pf.destroy()
    pd, cd, sd, ld = _evaluate(gpu, case, attrs, monkeypatch, "2", ti_mfma="1")   # the default
    pd.destroy()
    for a, ad, x, xd in zip(cf, cd, sf, sd):
        assert clv_ok(ad, a, exact=False) and (x == xd).all()
    pl, cl, sl, ll = _evaluate(gpu, case, attrs, monkeypatch, "0")
    o = oracle_run(orc, gpu, pl, case, attrs)
    o.update_partials()
    pl.destroy()
    for op, a, b, x, y in zip(plan.ops, cf, cl, sf, sl):
        node = int(op["parent_clv_index"])
        assert bits_equal(a, b), "CLV %d: whole list != per level" % node
        assert bits_equal(a, o.clv[node]), "CLV %d != oracle" % node
        assert (x == y).all() and (x == o.scalers[int(op["parent_scaler_index"])]).all()
    assert lf == ll


def test_whole_list_without_scale_buffers(gpu, orc, monkeypatch):
    case = _case(gpu, "balanced", 32, 700)
    plan = case["plan"]
    ops = plan.ops.copy()
    for f in ("parent_scaler_index", "child1_scaler_index", "child2_scaler_index"):
        ops[f] = SCALE_BUFFER_NONE
    res = {}
    for fused in ("2", "0"):
        monkeypatch.setenv("PLLHIP_AA_EXACT", "0")
        monkeypatch.setenv("PLLHIP_AA_CHERRY", "2")
        monkeypatch.setenv("PLLHIP_FUSED", fused)
        p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
        p.update_partials(ops)
        res[fused] = [p.get_clv(int(op["parent_clv_index"])) for op in ops]
        p.destroy()
    for a, b in zip(res["2"], res["0"]):
        assert bits_equal(a, b)


@pytest.mark.parametrize("attrs", [ATTRIB_PATTERN_TIP, 0])
@pytest.mark.parametrize("fused", ["0", "2"])
def test_many_tiles_per_wave_on_a_tree_that_scales(gpu, orc, monkeypatch, attrs, fused):
    """One workgroup walks the whole partition (PLLHIP_AA_GRID_CAP=1: 4+ tiles per wave) on a
    300-tip caterpillar, whose scaler counts reach 4: every count must equal the oracle's -- the
    per-level kernel read a wave's third and later tiles' inherited counts at the wrong sites until
    round 3 -- and every inner-inner CLV its bits."""
    monkeypatch.setenv("PLLHIP_AA_GRID_CAP", "1")
    case = _case(gpu, "caterpillar", 300, 300, seed=5)
    plan = case["plan"]
    for ti_mfma in ("0", "1"):
        p, clvs, scs, lnl = _evaluate(gpu, case, attrs, monkeypatch, fused, partial=False, ti_mfma=ti_mfma)
        o = oracle_run(orc, gpu, p, case, attrs)
        o.update_partials()
        assert p.scaling_certificate()["uncertified"] == 0
        p.destroy()
        top = 0
        kinds = _kinds(plan, 300, attrs)
        for op, kind, a, x in zip(plan.ops, kinds, clvs, scs):
            sc = o.scalers[int(op["parent_scaler_index"])]
            assert (x == sc).all(), "scaler counts of CLV %d" % int(op["parent_clv_index"])
            top = max(top, int(sc.max()))
            assert clv_ok(a, o.clv[int(op["parent_clv_index"])], exact=ti_mfma == "0"), kind
        assert top >= 3, "the tree was meant to scale (highest count %d)" % top
        ref = o.edge_loglikelihood(*plan.root_edge)
        assert abs(lnl - ref) <= 1e-11 * abs(ref)


def test_default_path_against_the_bit_exact_kernels_at_size(gpu, monkeypatch):
    """100,000 sites x 200 taxa (a random tree: tip-inner ops, evictions, three tiles per wave of the
    per-level kernel, scaling events near the root): scaler counts of the default path, whole list and
    per level, equal those of the bit-exact vector kernels; lnL within the stated tolerance."""
    from libpll_amd import workload as W
    T, sites, R = 200, 100_000, 4
    plan = W.random_tree(T, seed=42)
    rates, freqs = gpu.aa_model("lg")
    seqs = W.simulated_alignment(plan, sites, rates, freqs, gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    out = {}
    for name, env in (("whole list", {"PLLHIP_FUSED": "1", "PLLHIP_AA_EXACT": "0"}),
                      ("per level", {"PLLHIP_FUSED": "0", "PLLHIP_AA_EXACT": "0"}),
                      ("bit-exact", {"PLLHIP_FUSED": "0", "PLLHIP_AA_EXACT": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        p = W.setup_partition(gpu, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
        p.update_partials(plan.ops)
        lnl = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
        out[name] = (lnl, [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops[-12:]])
        p.destroy()
    assert max(int(s.max()) for s in out["bit-exact"][1]) >= 1, "no scaling event: the test has lost its point"
    for name in ("whole list", "per level"):
        for a, b in zip(out[name][1], out["bit-exact"][1]):
            assert (a == b).all(), name
        assert abs(out[name][0] - out["bit-exact"][0]) <= 1e-11 * abs(out["bit-exact"][0]), name


def test_whole_list_repeats_itself_at_size(gpu, monkeypatch):
    """100,000 sites x 200 taxa, the whole-list kernel eight times on fresh and on used partitions: every evaluation's
    per-site lnL equals the per-level launches' bit for bit.  (Round 3: an inner-inner op behind a run of barrier-free
    lookups read its left matrix block before every wave of the workgroup had staged its part -- one wave's tile
    wrong in one evaluation of six, and only at sizes where the waves have tiles enough to drift apart.)"""
    from libpll_amd import workload as W
    T, sites, R = 200, 100_000, 4
    plan = W.random_tree(T, seed=42)
    rates, freqs = gpu.aa_model("lg")
    seqs = W.simulated_alignment(plan, sites, rates, freqs, gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    monkeypatch.setenv("PLLHIP_AA_EXACT", "0")
    monkeypatch.setenv("PLLHIP_FUSED", "0")
    p = W.setup_partition(gpu, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
    p.update_partials(plan.ops)
    ref = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
    p.destroy()
    monkeypatch.setenv("PLLHIP_FUSED", "1")
    for fresh in range(2):
        # (fresh 0: tip-inner mat-vecs in the reference's order -- the per-level launches' bits; fresh 1: the default, on
        # the matrix cores -- the same value every time, and the per-level launches' to rounding)
        monkeypatch.setenv("PLLHIP_AA_TI_MFMA", str(fresh))
        p = W.setup_partition(gpu, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
        first = None
        for again in range(4):
            p.update_partials(plan.ops)
            got = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
            if fresh == 0:
                assert got[0] == ref[0] and bits_equal(got[1], ref[1]), "partition %d, evaluation %d" % (fresh, again)
            else:
                first = first or got
                assert got[0] == first[0] and bits_equal(got[1], first[1]), "evaluation %d differs from the first" % again
                assert abs(got[0] - ref[0]) <= 1e-12 * abs(ref[0]) and np.abs(got[1] - ref[1]).max() <= 1e-9
        assert p.scaling_certificate()["uncertified"] == 0
        p.destroy()


def test_tip_inner_on_the_matrix_cores_against_the_reference_order(gpu, orc, monkeypatch):
    """PLLHIP_AA_TI_MFMA (round 5: opt-in; round 6: the default, behind the scaling certificate): the ONE mat-vec of a
    tip-inner op of the whole-list kernel on the matrix cores (fused chains) instead of the vector unit in the
    reference's non-fused order (core_partials_avx.c:1229-1284).  Tip-inner CLVs -- and everything above them --
    agree with the reference-order path's (= the reference's bits) to 1e-13 relative on a 150-tip ladder (errors of
    ~1e-16 per op, carried up the tree), the scale buffers are equal (the certificate), lnL to 1e-12."""
    from helpers import make_case, build_partition, bits_equal, rel_err
    from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)
    monkeypatch.setenv("PLLHIP_FUSED", "2")
    for shape, tips, sites in (("caterpillar", 150, 700), ("random", 60, 2100)):
        case = make_case(20, shape, tips, sites, seed=tips)
        case["rates"], case["freqs"] = gpu.aa_model("lg")
        plan, R = case["plan"], case["rate_cats"]
        out = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("PLLHIP_AA_TI_MFMA", flag)
            p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
            p.update_partials(plan.ops)
            lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
            clvs = [p.get_clv(int(op["parent_clv_index"])) for op in plan.ops[::7]]
            scs = [p.get_scaler(int(op["parent_scaler_index"])) for op in plan.ops]
            out[flag] = (lnl, ps, clvs, scs)
            p.destroy()
        a, b = out["0"], out["1"]
        assert any(not bits_equal(x, y) for x, y in zip(a[2], b[2])), "the matrix-core path should differ in the last bits"
        for x, y in zip(a[2], b[2]):
            assert clv_ok(y, x, exact=False, tol=1e-13)
        for x, y in zip(a[3], b[3]):
            assert (x == y).all()
        assert rel_err(b[1], a[1]) < 1e-11 and abs(b[0] - a[0]) <= 1e-12 * abs(a[0])
