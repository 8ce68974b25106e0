"""The array-level entry points pll_core_* (reference: src/pll.h:827-1013,1659) through the
product's C-ABI: a whole tree is evaluated the way the reference's partials.c / likelihood.c /
derivatives.c drive their kernels -- op by op, every operand a host array -- and every
intermediate is compared with the oracle: P-matrices, CLVs and scale buffers bit for bit,
per-site lnL to 1e-13, lnL 1e-12, sumtable 1e-12, derivatives 1e-10.  4 and 20 states (and 5:
the generic kernels), pattern tips and tip CLVs, per-site and per-rate scalers."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_case, odd_state_case, build_partition, oracle_run, bits_equal, rel_err, sumtable_err
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, ATTRIB_ARCH_AVX2

pytestmark = pytest.mark.gpu

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_ubyte)


def d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def u(a):
    return None if a is None else a.ctypes.data_as(_up)


def b(a):
    return None if a is None else a.ctypes.data_as(_bp)


def rows(arrs):
    arr = (_dp * len(arrs))()
    for k, a in enumerate(arrs):
        arr[k] = a.ctypes.data_as(_dp)
    return arr


def bind(lib):
    L = lib.lib
    L.pll_core_edge_loglikelihood_ii.restype = C.c_double
    L.pll_core_edge_loglikelihood_ti.restype = C.c_double
    L.pll_core_edge_loglikelihood_ti_4x4.restype = C.c_double
    L.pll_core_root_loglikelihood.restype = C.c_double
    for name in ("pll_core_create_lookup", "pll_core_update_partial_tt", "pll_core_update_partial_ti",
                 "pll_core_update_partial_ii", "pll_amd_core_release"):
        getattr(L, name).restype = None
    return L


@pytest.mark.parametrize("states,shape,tips,sites,pattern_tip,rate_scalers",
                         [(4, "random", 12, 333, True, False), (4, "balanced", 8, 257, True, True),
                          (4, "random", 10, 200, False, False), (4, "caterpillar", 150, 64, True, False),
                          (20, "random", 9, 150, True, False), (20, "balanced", 8, 97, False, True),
                          (5, "random", 9, 120, True, False)])
def test_tree_through_core_api(gpu, orc, monkeypatch, states, shape, tips, sites, pattern_tip, rate_scalers):
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)   # (the default path: 20 states on the matrix cores)
    L = bind(gpu)
    if states in (4, 20):
        case = make_case(states, shape, tips, sites, seed=tips * 7 + sites)
    else:
        case = odd_state_case(states, tips=tips, sites=sites, seed=9, shape=shape)
    plan, S, R = case["plan"], states, case["rate_cats"]
    attrs = (ATTRIB_PATTERN_TIP if pattern_tip else 0) | (ATTRIB_RATE_SCALERS if rate_scalers else 0)
    # (an ISA bit is accepted and ignored where the reference's padding for it is none; 5 states under AVX2
    # would mean rows padded to 8 in the reference: refused, test_padded_layouts_are_refused)
    attrib = attrs | (ATTRIB_ARCH_AVX2 if states % 4 == 0 else 0)
    # a partition only to get the encoded tips and the host eigen system, and the oracle beside it
    p = build_partition(gpu, case, attrs)
    o = oracle_run(orc, gpu, p, case, attrs)
    o.update_partials()
    vals, vecs, inv = p.get_eigen(0)
    rates = gpu.compute_gamma_cats(case["alpha"], R)
    freqs = np.ascontiguousarray(o.m["freqs"], dtype=np.float64)
    vals, vecs, inv = (np.ascontiguousarray(x) for x in (vals, vecs, inv))
    span = S * R
    sc_len = sites * (R if rate_scalers else 1)

    # ---- P-matrices: pll_core_update_pmatrix (core_pmatrix.c:24)
    nm = plan.prob_matrices
    pm = np.zeros((nm, R, S, S))
    pm_rows = rows([pm[i] for i in range(nm)])
    mi = np.ascontiguousarray(plan.matrix_indices, dtype=np.uint32)
    bl = np.ascontiguousarray(plan.branch_lengths, dtype=np.float64)
    pi = np.zeros(R, dtype=np.uint32)
    pinv = np.zeros(1)
    assert L.pll_core_update_pmatrix(pm_rows, S, R, d(rates), d(bl), u(mi), u(pi), d(pinv), rows([vals]),
                                     rows([vecs]), rows([inv]), len(mi), attrib) == 1, gpu.errmsg()
    for m in plan.matrix_indices:
        assert bits_equal(pm[int(m)], o.pmat[int(m)]), "P-matrix %d" % m

    # ---- the op loop of partials.c:177-213 over host arrays
    nodes = 2 * tips - 2
    clv = np.zeros((nodes, sites, R, S))
    if not pattern_tip:
        clv[:tips] = o.clv[:tips]
    codes = o.tipcodes
    tipmap = o.tipmap
    maxstates = int(p.s.maxstates) if pattern_tip else 0
    scal = np.zeros((max(plan.scale_buffers, 1), sc_len), dtype=np.uint32)
    shift = 4 if S == 4 else int(np.ceil(np.log2(max(maxstates, 1))))
    table_rows = 256 if S == 4 else (((maxstates - 1) << shift) + maxstates if maxstates else 0)

    def sc(idx):
        return None if idx < 0 else scal[idx]

    for op in plan.ops:
        par, psc = int(op["parent_clv_index"]), int(op["parent_scaler_index"])
        c1, c2 = int(op["child1_clv_index"]), int(op["child2_clv_index"])
        m1, m2 = int(op["child1_matrix_index"]), int(op["child2_matrix_index"])
        s1, s2 = int(op["child1_scaler_index"]), int(op["child2_scaler_index"])
        t1, t2 = pattern_tip and c1 < tips, pattern_tip and c2 < tips
        if t1 and t2:
            lookup = np.zeros(table_rows * span)
            L.pll_core_create_lookup(S, R, d(lookup), d(pm[m1]), d(pm[m2]), u(tipmap), maxstates, attrib)
            L.pll_core_update_partial_tt(S, sites, R, d(clv[par]), u(sc(psc)), b(codes[c1]), b(codes[c2]),
                                         u(tipmap), maxstates, d(lookup), attrib)
        elif t1 or t2:
            tip, inner = (c1, c2) if t1 else (c2, c1)
            mt, mn = (m1, m2) if t1 else (m2, m1)
            sn = s2 if t1 else s1
            L.pll_core_update_partial_ti(S, sites, R, d(clv[par]), u(sc(psc)), b(codes[tip]), d(clv[inner]),
                                         d(pm[mt]), d(pm[mn]), u(sc(sn)), u(tipmap), maxstates, attrib)
        else:
            L.pll_core_update_partial_ii(S, sites, R, d(clv[par]), u(sc(psc)), d(clv[c1]), d(clv[c2]),
                                         d(pm[m1]), d(pm[m2]), u(sc(s1)), u(sc(s2)), attrib)
        assert bits_equal(clv[par], o.clv[par]), "CLV %d" % par
        if psc >= 0:
            assert (scal[psc] == o.scalers[psc]).all(), "scaler %d" % psc

    # ---- edge log-likelihood at the root edge (likelihood.c:416-513)
    pc, ps_, cc, cs, m = plan.root_edge
    pw = np.ascontiguousarray(o.pw, dtype=np.uint32)
    w = np.full(R, 1.0 / R)
    fi = np.zeros(R, dtype=np.uint32)
    persite = np.zeros(sites)
    if pattern_tip and (pc < tips or cc < tips):
        tip, inner, isc = (pc, cc, cs) if pc < tips else (cc, pc, ps_)
        lnl = L.pll_core_edge_loglikelihood_ti(S, sites, R, d(clv[inner]), u(sc(isc)), b(codes[tip]), u(tipmap),
                                               maxstates, d(pm[m]), rows([freqs]), d(w), u(pw), d(pinv), None,
                                               u(fi), d(persite), attrib)
    else:
        lnl = L.pll_core_edge_loglikelihood_ii(S, sites, R, d(clv[pc]), u(sc(ps_)), d(clv[cc]), u(sc(cs)),
                                               d(pm[m]), rows([freqs]), d(w), u(pw), d(pinv), None, u(fi),
                                               d(persite), attrib)
    want, want_ps = o.edge_loglikelihood(*plan.root_edge, persite=True)
    assert rel_err(persite, want_ps) < 1e-13
    assert abs(lnl - want) <= 1e-12 * abs(want)
    # ... and the root form on the last parent (likelihood.c:121, per-site scalers only)
    top = plan.ops[-1]
    tclv, tsc = int(top["parent_clv_index"]), int(top["parent_scaler_index"])
    if not rate_scalers:
        ra = L.pll_core_root_loglikelihood(S, sites, R, d(clv[tclv]), u(sc(tsc)), rows([freqs]), d(w), u(pw),
                                           d(pinv), None, u(fi), None, attrib)
        rb = p.update_partials(plan.ops) or p.compute_root_loglikelihood(tclv, tsc, [0] * R)
        assert abs(ra - rb) <= 1e-12 * abs(rb)

    # ---- sumtable and derivatives at the root edge (derivatives.c:164-312)
    st = np.zeros(sites * span)
    per_cat = lambda x: rows([x] * R)  # noqa: E731  (one model for all categories)
    if pattern_tip and (pc < tips or cc < tips):
        tip, inner, isc = (pc, cc, cs) if pc < tips else (cc, pc, ps_)
        ok = L.pll_core_update_sumtable_ti(S, sites, R, d(clv[inner]), b(codes[tip]), u(sc(isc)), per_cat(vecs),
                                           per_cat(inv), per_cat(freqs), u(tipmap), maxstates, d(st), attrib)
    else:
        ok = L.pll_core_update_sumtable_ii(S, sites, R, d(clv[pc]), d(clv[cc]), u(sc(ps_)), u(sc(cs)),
                                           per_cat(vecs), per_cat(inv), per_cat(freqs), d(st), attrib)
    assert ok == 1, gpu.errmsg()
    want_st = o.sumtable(pc, cc, ps_, cs)
    assert sumtable_err(st.reshape(want_st.shape), want_st) < 1e-12
    df, ddf = C.c_double(), C.c_double()
    pinv_cat = np.zeros(R)
    for t in (0.03, 0.4):
        assert L.pll_core_likelihood_derivatives(S, sites, R, d(w), u(sc(ps_)), u(sc(cs)), None, u(pw),
                                                 C.c_double(t), d(pinv_cat), per_cat(freqs), d(rates),
                                                 per_cat(vals), d(st), C.byref(df), C.byref(ddf), attrib) == 1
        assert rel_err(np.array([df.value, ddf.value]), np.array(o.derivatives(want_st, t))) < 1e-10
    p.destroy()
    L.pll_amd_core_release()


@pytest.mark.parametrize("states,flag,padded", [(5, "avx", 8), (7, "avx2", 8), (5, "sse", 6), (7, "sse", 8), (61, "avx2", 64)])
def test_padded_layouts_are_refused(gpu, states, flag, padded):
    """The reference pads rows of states under its SIMD flags (pll.c:437-451); this library takes
    unpadded arrays.  A call that carries such a flag with a state count the flag would pad is
    refused with PLL_ERROR_PARAM_INVALID instead of being read with the wrong stride; without the
    flag the same call works."""
    from libpll_amd.pllapi import ATTRIB_ARCH_AVX, ATTRIB_ARCH_SSE, ERROR_PARAM_INVALID
    L = bind(gpu)
    bit = {"avx": ATTRIB_ARCH_AVX, "avx2": ATTRIB_ARCH_AVX2, "sse": ATTRIB_ARCH_SSE}[flag]
    sites, R = 17, 2
    rng = np.random.default_rng(3)
    left, right = rng.random((sites, R, states)), rng.random((sites, R, states))
    pm = rng.random((2, R, states, states))
    parent = np.full((sites, R, states), -1.0)
    gpu.clear_error()
    L.pll_core_update_partial_ii(states, sites, R, d(parent), None, d(left), d(right), d(pm[0]), d(pm[1]), None, None, bit)
    assert gpu.errno() == ERROR_PARAM_INVALID and str(padded) in gpu.errmsg()
    assert (parent == -1.0).all(), "nothing may have been written"
    gpu.clear_error()
    L.pll_core_update_partial_ii(states, sites, R, d(parent), None, d(left), d(right), d(pm[0]), d(pm[1]), None, None, 0)
    assert gpu.errno() == 0, gpu.errmsg()
    want = np.einsum("kij,nkj->nki", pm[0], left) * np.einsum("kij,nkj->nki", pm[1], right)
    assert np.allclose(parent, want, rtol=1e-13)
    L.pll_amd_core_release()


def test_lookup_rows_of_character_zero_are_zero(gpu):
    """pll_core_create_lookup for 4 states fills all 256 rows of the caller's table: those of
    character 0, which no sequence contains, with zeros (core_partials.c:665-723) -- they used to
    be left as they were, and pll_core_update_partial_tt uploads all 256."""
    L = bind(gpu)
    R = 4
    rng = np.random.default_rng(5)
    pm = rng.random((2, R, 4, 4))
    lookup = np.full(256 * 4 * R, np.nan)
    L.pll_core_create_lookup(4, R, d(lookup), d(pm[0]), d(pm[1]), None, 0, 0)
    t = lookup.reshape(16, 16, R * 4)
    assert (t[0] == 0.0).all() and (t[:, 0] == 0.0).all()
    assert np.isfinite(t[1:, 1:]).all() and (t[1:, 1:] > 0).all()
    L.pll_amd_core_release()


def test_core_api_keeps_a_context_per_recent_shape(gpu):
    """Alternating shapes (what a traversal through pll_core_update_partial_tt/ti/ii does) must not
    rebuild the scratch context call after call: the second round of the same shapes is much faster
    than the first."""
    import time
    L = bind(gpu)
    L.pll_amd_core_release()
    rng = np.random.default_rng(1)
    shapes = [(4, 100, 4), (20, 60, 2), (4, 300, 1)]
    data = []
    for S, sites, R in shapes:
        data.append((rng.random((sites, R, S)), rng.random((sites, R, S)), rng.random((2, R, S, S)), np.zeros((sites, R, S))))

    def round_():
        t0 = time.perf_counter()
        for (S, sites, R), (l, r, pm, par) in zip(shapes, data):
            L.pll_core_update_partial_ii(S, sites, R, d(par), None, d(l), d(r), d(pm[0]), d(pm[1]), None, None, 0)
        return time.perf_counter() - t0
    first = round_()
    later = min(round_() for _ in range(3))
    assert later < 0.5 * first, (first, later)
    L.pll_amd_core_release()
