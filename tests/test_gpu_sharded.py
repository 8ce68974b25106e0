"""One partition over several devices of one process (pll_amd_set_devices / PLL_AMD_DEVICES;
libpll_amd/csrc/hip/shard.hip): an unmodified client -- pll_partition_create (pll.h:530),
pll_update_partials, pll_compute_edge_loglikelihood -- gets its sites split over the devices.

A device ordinal may repeat, so the whole mechanism (split uploads, fan-out of every call,
gathered mirrors, host sum of the per-shard results, the ascertainment-bias sites on the last
shard) runs on a one-GPU box with "0,0" and "0,0,0"; with two or more devices visible the same
checks run over distinct devices as well.  Sharded results must equal the unsharded ones:
CLVs, scale buffers, sumtables and per-site lnL bit for bit, sums to 1e-12 (the summation tree
differs), and the oracle's within the usual tolerances."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_case, build_partition, oracle_run, bits_equal, rel_err
from libpll_amd import workload as W
from libpll_amd.pllapi import (ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, ATTRIB_SITE_REPEATS,
                               ATTRIB_AB_LEWIS, ATTRIB_AB_STAMATAKIS, PllError)

pytestmark = pytest.mark.gpu


class devices:
    """with devices(lib, [0, 0]): partitions created inside are sharded over that list."""

    def __init__(self, lib, devs):
        self.lib, self.devs = lib, devs

    def __enter__(self):
        arr = (C.c_int * max(1, len(self.devs)))(*self.devs)
        assert self.lib.lib.pll_amd_set_devices(arr, len(self.devs)) == 1

    def __exit__(self, *exc):
        self.lib.lib.pll_amd_set_devices(None, 0)


def device_lists(gpu):
    lists = [[0, 0], [0, 0, 0]]
    n = gpu.device_count()
    if n >= 2:
        lists.append(list(range(min(n, 8))))
    return lists


def full_state(p, plan, R, pinv_sites=False):
    """Everything a client can observe after one evaluation + the Newton leg."""
    p.update_partials(plan.ops)
    e = plan.root_edge
    lnl, ps = p.compute_edge_loglikelihood(*e, [0] * R, persite=True)
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    table = p.get_sumtable(st)
    d = [p.compute_likelihood_derivatives(e[1], e[3], t, [0] * R, st) for t in (0.04, 0.3)]
    clvs = {int(op["parent_clv_index"]): p.get_clv(int(op["parent_clv_index"])) for op in plan.ops}
    scs = {int(op["parent_scaler_index"]): p.get_scaler(int(op["parent_scaler_index"]))
           for op in plan.ops if int(op["parent_scaler_index"]) >= 0}
    top = plan.ops[-1]
    root = p.compute_root_loglikelihood(int(top["parent_clv_index"]), int(top["parent_scaler_index"]), [0] * R)
    return dict(lnl=lnl, ps=ps, table=table, d=d, clvs=clvs, scs=scs, root=root)


def assert_same(a, b):
    assert bits_equal(a["ps"], b["ps"]), "per-site lnL"
    assert abs(a["lnl"] - b["lnl"]) <= 1e-12 * abs(b["lnl"])
    assert abs(a["root"] - b["root"]) <= 1e-12 * abs(b["root"])
    assert bits_equal(a["table"], b["table"]), "sumtable"
    assert rel_err(np.array(a["d"]), np.array(b["d"])) < 1e-11
    for node in b["clvs"]:
        assert bits_equal(a["clvs"][node], b["clvs"][node]), "CLV %d" % node
    for sc in b["scs"]:
        assert (a["scs"][sc] == b["scs"][sc]).all(), "scaler %d" % sc


@pytest.mark.parametrize("states,shape,tips,sites,attrs,pinv,rate_cats",
                         [(4, "random", 20, 3001, ATTRIB_PATTERN_TIP, 0.0, 4),
                          (4, "balanced", 16, 1000, ATTRIB_PATTERN_TIP | ATTRIB_RATE_SCALERS, 0.0, 4),
                          (4, "caterpillar", 300, 700, ATTRIB_PATTERN_TIP, 0.0, 4),
                          # (per-rate scale buffers on a tree that scales: the root kernel reads ENTRY i of the
                          # buffer for site i, core_likelihood.c:197-198 -- entries other shards hold)
                          (4, "caterpillar", 300, 700, ATTRIB_PATTERN_TIP | ATTRIB_RATE_SCALERS, 0.0, 4),
                          (4, "random", 12, 2500, 0, 0.0, 4),
                          (4, "random", 14, 1500, ATTRIB_PATTERN_TIP, 0.2, 4),
                          (20, "random", 12, 900, ATTRIB_PATTERN_TIP, 0.0, 4),
                          (20, "balanced", 8, 777, ATTRIB_RATE_SCALERS, 0.0, 4),
                          # (20 states x 6 / 8 categories: every op in chunks of the categories, round 4)
                          (20, "random", 12, 900, ATTRIB_PATTERN_TIP, 0.1, 6),
                          (20, "caterpillar", 120, 600, ATTRIB_RATE_SCALERS, 0.0, 8),
                          (7, "random", 9, 1100, ATTRIB_PATTERN_TIP, 0.0, 4)])
def test_sharded_equals_unsharded(gpu, orc, monkeypatch, states, shape, tips, sites, attrs, pinv, rate_cats):
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)   # (the default path: 20 states on the matrix cores)
    monkeypatch.setenv("PLLHIP_FUSED", "2")     # 4 states: the whole-list kernel also on these small shards
    if states in (4, 20):
        case = make_case(states, shape, tips, sites, seed=tips + sites, rate_cats=rate_cats)
    else:
        from helpers import odd_state_case
        case = odd_state_case(states, tips=tips, sites=sites, seed=5)
    plan, R = case["plan"], case["rate_cats"]
    whole_p = build_partition(gpu, case, attrs, pinv=pinv)
    assert gpu.lib.pll_amd_shard_count(whole_p.ptr) == 1
    o = oracle_run(orc, gpu, whole_p, case, attrs, pinv=pinv)
    whole = full_state(whole_p, plan, R)
    o.update_partials()
    ref_lnl = o.edge_loglikelihood(*plan.root_edge)
    assert abs(whole["lnl"] - ref_lnl) <= 1e-12 * abs(ref_lnl)
    whole_p.destroy()
    for devs in device_lists(gpu):
        with devices(gpu, devs):
            p = build_partition(gpu, case, attrs, pinv=pinv)
        per = -(-(-(-sites // len(devs))) // 256) * 256          # ceil(ceil(sites / n) / 256) * 256
        assert gpu.lib.pll_amd_shard_count(p.ptr) == -(-sites // per)
        got = full_state(p, plan, R)
        assert_same(got, whole)
        assert abs(got["lnl"] - ref_lnl) <= 1e-12 * abs(ref_lnl)
        # a branch-length change and the partial traversal that follows it
        changed = int(plan.ops[0]["parent_clv_index"])
        p.update_prob_matrices([0] * R, [changed], [0.41])
        p.update_partials(plan.ops)
        again = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
        assert again != got["lnl"]
        p.destroy()


def test_small_partitions_get_fewer_shards(gpu):
    """Ranges are multiples of 256 sites: 300 sites over three devices are two ranges, 200
    sites one -- and the results do not care."""
    for sites, expect in ((300, 2), (200, 1), (513, 3)):
        case = make_case(4, "random", 8, sites, seed=sites)
        with devices(gpu, [0, 0, 0]):
            p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
        assert gpu.lib.pll_amd_shard_count(p.ptr) == expect
        q = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
        for x in (p, q):
            x.update_partials(case["plan"].ops)
        a = p.compute_edge_loglikelihood(*case["plan"].root_edge, [0] * 4, persite=True)
        b = q.compute_edge_loglikelihood(*case["plan"].root_edge, [0] * 4, persite=True)
        assert bits_equal(a[1], b[1]) and abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
        p.destroy()
        q.destroy()


def test_per_shard_stopwatch(gpu):
    """pll_amd_timer_shard_ms: the stopwatch of a sharded partition reports the slowest shard; the per-shard
    figures (what bench.py prints as per_rank_ms_per_step in --in-process mode) show every device's own time."""
    case = make_case(4, "balanced", 16, 40_000, seed=3)
    with devices(gpu, [0, 0, 0]):
        p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    assert gpu.lib.pll_amd_shard_count(p.ptr) == 3
    p.update_partials(case["plan"].ops)
    p.wait()
    p.timer_start()
    for _ in range(5):
        p.update_partials(case["plan"].ops)
    slowest = p.timer_stop_ms()
    per = p.shard_ms()
    assert len(per) == 3 and all(t > 0.0 for t in per)
    assert abs(max(per) - slowest) <= 1e-6 * slowest
    p.destroy()
    q = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    q.timer_start()
    q.update_partials(case["plan"].ops)
    t = q.timer_stop_ms()
    assert q.shard_ms() == [pytest.approx(t)]
    q.destroy()


@pytest.mark.parametrize("kind", [ATTRIB_AB_LEWIS, ATTRIB_AB_STAMATAKIS])
def test_sharded_ascertainment_bias(gpu, kind):
    """The per-state sites and their correction live on the last shard; the value is the same."""
    from test_gpu_asc_bias import build
    case = make_case(4, "random", 12, 1300, seed=77)
    plan, R = case["plan"], case["rate_cats"]
    sw = np.arange(1, 5, dtype=np.uint32) * 3 if kind == ATTRIB_AB_STAMATAKIS else None
    whole = build(gpu, case, kind | ATTRIB_PATTERN_TIP, sw)
    with devices(gpu, [0, 0, 0]):
        split = build(gpu, case, kind | ATTRIB_PATTERN_TIP, sw)
    assert gpu.lib.pll_amd_shard_count(split.ptr) == 3
    out = []
    for p in (whole, split):
        p.update_partials(plan.ops)
        e = plan.root_edge
        lnl = p.compute_edge_loglikelihood(*e, [0] * R)
        st = p.alloc_sumtable()
        p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
        d = p.compute_likelihood_derivatives(e[1], e[3], 0.2, [0] * R, st)
        top = int(plan.ops[-1]["parent_clv_index"])
        out.append((lnl, d, p.get_clv(top)))
        p.destroy()
    assert abs(out[0][0] - out[1][0]) <= 1e-12 * abs(out[0][0])
    assert rel_err(np.array(out[1][1]), np.array(out[0][1])) < 1e-11
    assert bits_equal(out[0][2], out[1][2])


def test_sharded_refusals(gpu):
    case = make_case(4, "balanced", 8, 600, seed=1)
    with devices(gpu, [0, 0]):
        # (site repeats over several devices: available since round 4, test_sharded_with_site_repeats)
        p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    with pytest.raises(PllError):
        p.comm_init(0, 1, b"\0" * 128)
    p.destroy()
    with devices(gpu, [0, 99]):
        with pytest.raises(PllError):
            build_partition(gpu, case, ATTRIB_PATTERN_TIP)


@pytest.mark.parametrize("threads", ["1", "0"])
def test_sharded_errors_come_back_from_the_shards_threads(gpu, monkeypatch, threads):
    """Round 5: the hot calls on a sharded partition are enqueued by one host thread per shard (shard.hip,
    pllhip_group_parallel).  An error raised on a shard's thread -- an op that names a CLV the partition does not have,
    a negative branch length -- comes back to the caller with its text (the library's error state is per thread), and the
    partition works on afterwards; the same with the calling thread visiting the shards in turn (PLLHIP_SHARD_THREADS=0)."""
    monkeypatch.setenv("PLLHIP_SHARD_THREADS", threads)
    case = make_case(4, "balanced", 8, 1500, seed=5)
    plan, R = case["plan"], case["rate_cats"]
    with devices(gpu, [0, 0, 0]):
        p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    assert gpu.lib.pll_amd_shard_count(p.ptr) == 3
    p.update_partials(plan.ops)
    want = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
    bad = plan.ops.copy()
    bad["child1_clv_index"][2] = 10_000
    gpu.clear_error()
    p.update_partials(bad)                            # (a void function: errors through pll_errno, as in the reference)
    assert gpu.errno() == 201 and "out of range" in gpu.errmsg(), (gpu.errno(), gpu.errmsg())   # PLL_ERROR_HIP_RUNTIME
    with pytest.raises(PllError):
        p.update_prob_matrices([0] * R, [0], [-1.0])
    for _ in range(20):                               # many calls in a row: the threads' hand-over
        p.update_partials(plan.ops)
    assert p.compute_edge_loglikelihood(*plan.root_edge, [0] * R) == want
    p.destroy()


def test_environment_device_list(gpu, monkeypatch):
    case = make_case(4, "balanced", 8, 1024, seed=2)
    monkeypatch.setenv("PLL_AMD_DEVICES", "0,0-0,0")      # three entries: ranges of 512 sites, so two shards
    p = build_partition(gpu, case, ATTRIB_PATTERN_TIP)
    assert gpu.lib.pll_amd_shard_count(p.ptr) == 2
    p.destroy()
    monkeypatch.setenv("PLL_AMD_DEVICES", "0;1")
    with pytest.raises(PllError):
        build_partition(gpu, case, ATTRIB_PATTERN_TIP)


def test_sharded_at_config_2_size(gpu):
    """BASELINE config 2's partition (1,000,000 sites, 64 taxa) in four ranges: per-site lnL
    bitwise equal to the unsharded partition's, lnL to 1e-12, and faster than nothing: the
    four ranges' op lists run side by side."""
    sites, T, R = 1_000_000, 64, 4
    plan = W.balanced_tree(T, seed=42)
    seqs = W.simulated_alignment(plan, sites, W.GTR_RATES, W.GTR_FREQS,
                                 gpu.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
    whole = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP)
    n = gpu.device_count()
    with devices(gpu, list(range(n)) if n >= 2 else [0, 0, 0, 0]):
        split = W.setup_partition(gpu, plan, seqs, 4, R, ATTRIB_PATTERN_TIP)
    assert gpu.lib.pll_amd_shard_count(split.ptr) == (n if n >= 2 else 4)
    res = []
    for p in (whole, split):
        p.update_partials(plan.ops)
        res.append(p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True))
        p.destroy()
    assert bits_equal(res[0][1], res[1][1])
    assert abs(res[0][0] - res[1][0]) <= 1e-12 * abs(res[0][0])


@pytest.mark.parametrize("states,shape,tips,sites,rate_scalers",
                         [(4, "random", 20, 3001, False), (4, "balanced", 16, 2000, True), (20, "random", 12, 900, False),
                          # (deep trees that SCALE: a shard stores a class-stored CLV's scale buffer by class too, and
                          # the root kernel of a per-rate partition reads entries of the WHOLE buffer -- ADVICE r4)
                          (4, "caterpillar", 300, 1500, True), (4, "caterpillar", 300, 1500, False),
                          (20, "caterpillar", 150, 800, True)])
def test_sharded_with_site_repeats(gpu, monkeypatch, states, shape, tips, sites, rate_scalers):
    """One partition over several devices WITH PLL_ATTRIB_SITE_REPEATS (round 4; BASELINE config 4's split and
    config 5's feature together): every shard identifies the classes of its own site range, and everything a
    client can observe -- expanded CLVs and scale buffers, per-site lnL, sumtable bitwise; lnL, derivatives --
    equals the unsharded PLAIN partition.  Repeated columns make most nodes stored by class on every shard."""
    monkeypatch.delenv("PLLHIP_AA_EXACT", raising=False)
    # (site repeats run per level, the plain partition may take the whole-list kernel: bit for bit only with its
    # tip-inner mat-vecs in the reference's order too -- the default's rounding is tests/test_gpu_cert.py's subject)
    monkeypatch.setenv("PLLHIP_AA_TI_MFMA", "0")
    case = make_case(states, shape, tips, sites, seed=tips + sites, gap_frac=0.02)
    rng = np.random.default_rng(sites)
    pool = rng.integers(0, sites, size=sites // 6 + 1)
    pick = pool[rng.integers(0, len(pool), size=sites)]
    case["seqs"] = [bytes(np.frombuffer(s, dtype=np.uint8)[pick]) for s in case["seqs"]]
    plan, R = case["plan"], case["rate_cats"]
    attrs = ATTRIB_PATTERN_TIP | (ATTRIB_RATE_SCALERS if rate_scalers else 0)
    plain = build_partition(gpu, case, attrs)
    whole = full_state(plain, plan, R)
    plain.destroy()
    for devs in device_lists(gpu):
        with devices(gpu, devs):
            p = build_partition(gpu, case, attrs | ATTRIB_SITE_REPEATS)
        assert gpu.lib.pll_amd_shard_count(p.ptr) > 1
        got = full_state(p, plan, R)
        assert_same(got, whole)
        if shape == "caterpillar":
            assert max(int(v.max()) for v in whole["scs"].values()) > 0, "this tree must scale"
        rows = [p.repeats_classes(int(op["parent_clv_index"])) for op in plan.ops]
        assert sum(1 for r in rows if 0 < r < sites) > len(plan.ops) // 2, "most nodes are stored by class"
        # a topology-neutral change (branch length) and a tip change, then again: still the plain partition's values
        changed = int(plan.ops[0]["parent_clv_index"])
        p.update_prob_matrices([0] * R, [changed], [0.41])
        q = build_partition(gpu, case, attrs)
        q.update_prob_matrices([0] * R, [changed], [0.41])
        for x in (p, q):
            x.update_partials(plan.ops)
        a = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
        b = q.compute_edge_loglikelihood(*plan.root_edge, [0] * R, persite=True)
        assert bits_equal(a[1], b[1]) and abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
        p.destroy()
        q.destroy()
