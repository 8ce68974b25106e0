"""BASELINE.json configs[3] and configs[4] at full size on one GPU.

C4: 4-state GTR, 4 rates, 8,000,000 sites, 128-taxon balanced tree (the whole alignment of the
    8-GPU job on ONE device: 126 CLVs x 1 GB = 133 GB of the 288 GB; every shard of the 8-GPU run
    executes this very 126-op list).
C5: 4-state GTR+G, 500,000 sites, 200-taxon random tree, plain and with PLL_ATTRIB_SITE_REPEATS,
    pll_update_sumtable + five pll_compute_likelihood_derivatives calls (the Newton inner loop,
    examples/newton/newton.c).

At these sizes the checks are the size-independent properties (reduction complete, sites
independent, run-to-run reproducible, repeats == plain) plus slices against the genuine reference
(oracle/_ref, AVX2 flag): scaler counts and CLVs bitwise, per-site lnL 1e-13, d/dd 1e-10.
Reference shapes: test/src/scaling.c:263-367 (deep tree + scalers + sumtable/derivatives)."""
import numpy as np
import pytest

from helpers import bits_equal, clv_ok, rel_err
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_ARCH_AVX2, ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, ATTRIB_SITE_REPEATS, PllError

pytestmark = pytest.mark.gpu

R = 4
FI = [0] * R
PERSITE_RTOL = 1e-13   # the device's log against the C library's: last-bit differences only


@pytest.fixture(scope="module")
def c4_block(gpu):
    """A 1,000,000-site block simulated down the 128-taxon tree; the 8 M-site alignment is
    eight copies of it (generation stays at seconds, and equal columns must give equal bits
    wherever they lie)."""
    plan = W.balanced_tree(128, seed=42)
    block = W.simulated_alignment(plan, 1_000_000, W.GTR_RATES, W.GTR_FREQS,
                                  gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    return plan, block


def test_config4_whole_alignment_on_one_gpu(gpu, c4_block):
    plan, block = c4_block
    copies, n = 8, len(block[0])
    sites = copies * n
    seqs = [s * copies for s in block]
    try:
        p = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP)
    except PllError as e:   # 133 GB of CLVs: an MI355X has them
        pytest.fail("8,000,000 sites x 128 taxa did not fit this device: %s" % e)
    p.update_partials(plan.ops)
    lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
    assert np.isfinite(lnl) and lnl < 0
    assert abs(ps.sum() - lnl) <= 1e-11 * abs(lnl)
    # equal columns, equal bits: every copy of the block, wherever its tiles lie
    blocks = ps.reshape(copies, n)
    for k in range(1, copies):
        assert bits_equal(blocks[k], blocks[0]), "copy %d of the block" % k
    top = int(plan.ops[-1]["parent_scaler_index"])
    top_scaler = p.get_scaler(top)
    assert (top_scaler.reshape(copies, n) == top_scaler[:n]).all()
    # again (the whole-list kernel walks the tiles the other way round on every other launch
    # of a partition this large): same bits
    p.update_partials(plan.ops)
    lnl2, ps2 = p.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
    assert lnl2 == lnl and bits_equal(ps2, ps)
    p.destroy()
    # halves: the site-sharding identity the multi-GPU path relies on, per site and bitwise
    total = 0.0
    for lo, hi in ((0, sites // 2), (sites // 2, sites)):
        h = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP, site_range=(lo, hi))
        h.update_partials(plan.ops)
        v, hps = h.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
        assert bits_equal(hps, ps[lo:hi])
        total += v
        h.destroy()
    assert abs(total - lnl) <= 1e-12 * abs(lnl)


@pytest.mark.parametrize("path", ["fused", "levels"])
def test_config4_slice_against_reference(gpu, ref, c4_block, monkeypatch, path):
    """The first 100,000 sites of the C4 alignment through the genuine reference and through
    both 4-state paths (the whole-list kernel forced: the 126-op list needs six slots per wave)."""
    monkeypatch.setenv("PLLHIP_FUSED", "2" if path == "fused" else "0")
    plan, block = c4_block
    seqs = [s[:100_000] for s in block]
    a = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP)
    r = W.setup_partition(ref, plan, seqs, 4, R, ATTRIB_PATTERN_TIP | ATTRIB_ARCH_AVX2)
    a.update_partials(plan.ops)
    r.update_partials(plan.ops)
    for op in list(plan.ops[-4:]) + list(plan.ops[70:73]):
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        assert (a.get_scaler(sc) == r.get_scaler(sc)).all(), "scaler %d" % sc
        assert bits_equal(a.get_clv(node), r.get_clv(node)), "CLV %d" % node
    la, pa = a.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
    lr, pr = r.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
    assert rel_err(pa, pr) < PERSITE_RTOL
    assert abs(la - lr) <= 1e-10 * abs(lr)
    a.destroy()
    r.destroy()


@pytest.fixture(scope="module")
def c5_data(gpu):
    plan = W.random_tree(200, seed=42)
    seqs = W.simulated_alignment(plan, 500_000, W.GTR_RATES, W.GTR_FREQS,
                                 gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    return plan, seqs


NEWTON_T = (0.02, 0.05, 0.11, 0.23, 0.5)


def newton_leg(p, plan):
    e = plan.root_edge
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], FI, st)
    return [p.compute_likelihood_derivatives(e[1], e[3], t, FI, st) for t in NEWTON_T]


def test_config5_plain_and_site_repeats(gpu, c5_data):
    plan, seqs = c5_data
    out = {}
    for name, extra in (("plain", 0), ("repeats", ATTRIB_SITE_REPEATS)):
        p = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP | extra)
        p.update_partials(plan.ops)
        lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
        assert np.isfinite(lnl) and lnl < 0 and abs(ps.sum() - lnl) <= 1e-11 * abs(lnl)
        d = newton_leg(p, plan)
        # a second evaluation (kept plan / kept classes) gives the same bits
        p.update_partials(plan.ops)
        lnl2, ps2 = p.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
        assert lnl2 == lnl and bits_equal(ps2, ps)
        sc = p.get_scaler(int(plan.ops[-1]["parent_scaler_index"]))
        if extra:
            rows = [p.repeats_classes(int(op["parent_clv_index"])) for op in plan.ops]
            assert sum(1 for c in rows if c) > len(plan.ops) // 2      # most nodes are stored by class
        out[name] = (lnl, ps, d, sc)
        p.destroy()
    a, b = out["plain"], out["repeats"]
    assert a[0] == b[0] and bits_equal(a[1], b[1]), "site repeats change the per-site lnL"
    assert a[2] == b[2], "site repeats change the derivatives"
    assert (a[3] == b[3]).all()
    # derivatives are those of -lnL: negative slope where lnL still rises, and they vary with t
    assert len({d for d, _ in a[2]}) == len(NEWTON_T)


@pytest.mark.parametrize("scale_attr", [0, ATTRIB_RATE_SCALERS], ids=["per-site-scalers", "per-rate-scalers"])
@pytest.mark.parametrize("repeats", [False, True])
def test_config5_slice_against_reference(gpu, ref, c5_data, repeats, scale_attr):
    """The first 50,000 sites of the C5 alignment: scaler counts and CLVs bitwise, per-site lnL to 1e-13
    against the genuine reference, the five derivative pairs to 1e-10 -- plain and with site
    repeats (expanded CLVs); with per-site and with per-rate scale buffers (the reference's per-rate
    rule, core_partials_avx.c:494-503, on a tree that scales)."""
    plan, full = c5_data
    seqs = [s[:50_000] for s in full]
    a = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP | scale_attr | (ATTRIB_SITE_REPEATS if repeats else 0))
    r = W.setup_partition(ref, plan, seqs, 4, R, ATTRIB_PATTERN_TIP | scale_attr | ATTRIB_ARCH_AVX2)
    a.update_partials(plan.ops)
    r.update_partials(plan.ops)
    for op in list(plan.ops[-5:]) + list(plan.ops[100:103]):
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        assert (a.get_scaler(sc) == r.get_scaler(sc)).all(), "scaler %d" % sc
        assert bits_equal(a.get_clv(node), r.get_clv(node)), "CLV %d" % node
    la, pa = a.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
    lr, pr = r.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
    assert rel_err(pa, pr) < PERSITE_RTOL
    assert abs(la - lr) <= 1e-10 * abs(lr)
    da, dr = newton_leg(a, plan), newton_leg(r, plan)
    assert rel_err(np.array(da), np.array(dr)) < 1e-10, (da, dr)
    a.destroy()
    r.destroy()


# ---- BASELINE config 3 (20-state LG, 4 rates, 64 taxa) and the same data on a 200-taxon random tree: slices
# against the GENUINE REFERENCE on the DEFAULT path (until round 4 every 20-state check that was
# tied to the reference itself ran the non-default vector kernels)
@pytest.fixture(scope="module", params=["balanced-64", "random-200"])
def c3_data(request, gpu):
    plan = W.balanced_tree(64, seed=42) if request.param == "balanced-64" else W.random_tree(200, seed=42)
    rates, freqs = gpu.aa_model("lg")
    seqs = W.simulated_alignment(plan, 50_000, rates, freqs, gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    return request.param, plan, seqs


@pytest.mark.parametrize("scale_attr", [0, ATTRIB_RATE_SCALERS], ids=["per-site-scalers", "per-rate-scalers"])
@pytest.mark.parametrize("path", ["whole-list", "whole-list-reference-order", "levels"])
def test_config3_slice_against_reference(gpu, ref, c3_data, monkeypatch, path, scale_attr):
    """50,000 sites of C3 -- and of an LG alignment on a 200-taxon random tree: tip-inner ops, scaling events,
    evictions -- through the genuine reference (AVX2 flag) and through the product's DEFAULT 20-state path, whole
    list and per level: the scale buffer of EVERY op bitwise, CLVs (every fifth op and the last five: a CLV is 32 MB)
    bitwise -- on the default whole-list path (round 6: tip-inner mat-vecs on the matrix cores behind the scaling
    certificate) to 1e-13 wherever a tip-inner op lies below; with PLLHIP_AA_TI_MFMA=0 bitwise everywhere --,
    per-site lnL to 1e-11 (the edge kernel on the matrix cores sums a row in one fused chain), lnL to 1e-12.
    Per-rate scale buffers (round 6) on the whole-list kernel too."""
    name, plan, seqs = c3_data
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)
    monkeypatch.setenv("PLLHIP_FUSED", "0" if path == "levels" else "2")
    monkeypatch.setenv("PLLHIP_AA_TI_MFMA", "0" if path == "whole-list-reference-order" else "1")
    exact = path != "whole-list"
    a = W.setup_partition(gpu, plan, seqs, 20, R, ATTRIB_PATTERN_TIP | scale_attr)
    r = W.setup_partition(ref, plan, seqs, 20, R, ATTRIB_PATTERN_TIP | scale_attr | ATTRIB_ARCH_AVX2)
    a.update_partials(plan.ops)
    a.update_partials(plan.ops)     # (the same list again: the whole-list plan with its tip-tip ops inside)
    r.update_partials(plan.ops)
    top = 0
    for i, op in enumerate(plan.ops):
        node, sc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        rs = r.get_scaler(sc)
        assert (a.get_scaler(sc) == rs).all(), "%s: scale buffer of op %d" % (name, i)
        top = max(top, int(rs.max()))
        if i % 5 == 0 or i >= len(plan.ops) - 5:
            assert clv_ok(a.get_clv(node), r.get_clv(node), exact, tol=1e-13), "%s: CLV of op %d" % (name, i)
    if name == "random-200":
        assert top >= 1, "the 200-taxon tree was meant to scale"
    cert = a.scaling_certificate()
    assert cert["uncertified"] == 0 and (cert["lists"] >= 1) == (path == "whole-list" and name == "random-200"), cert
    la, pa = a.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
    lr, pr = r.compute_edge_loglikelihood(*plan.root_edge, FI, persite=True)
    assert rel_err(pa, pr) < 1e-11
    assert abs(la - lr) <= 1e-12 * abs(lr)
    a.destroy()
    r.destroy()
