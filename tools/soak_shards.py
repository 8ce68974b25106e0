#!/usr/bin/env python3
"""developer tool (round 5): partitions sharded over 2-8 "devices" (ordinal 0 repeated) created, used and destroyed in
a loop -- the group's enqueueing threads (shard.hip: one per shard, started on first use, joined by the destroy), the
polled results, errors raised on a shard's thread -- against the unsharded partition's values: per-site lnL bitwise,
lnL and derivatives to 1e-12.  Looks for hangs, crashes and differences, not for speed.
  python3 tools/soak_shards.py [first seed] [count] [shards: 0 = 2-8 at random] [seconds: stop after that long]
Round 6: eight shards every time, for ten minutes, with fewer cores than threads --
  taskset -c 0,1 python3 tools/soak_shards.py 0 1000000 8 600
-- is what shows a lost wake-up or a starved pool (a shard's thread spins 2,000 times, then sleeps on a condition
variable; the caller spins 20,000 times, then yields)."""
import ctypes as C
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import libpll_amd
from helpers import make_case, build_partition, bits_equal, rel_err
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, ATTRIB_SITE_REPEATS

# (per-site lnL is compared bit for bit with the unsharded partition's: a shard is smaller than the whole and may take
# the per-level launches where the whole takes the 20-state whole-list kernel -- the same bits only with that kernel's
# tip-inner mat-vecs in the reference's order; the default's rounding is tests/test_gpu_cert.py's subject)
os.environ["PLLHIP_AA_TI_MFMA"] = "0"
amd = libpll_amd.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fixed_k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
budget_s = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
done = 0
bad, t0 = 0, time.time()


def observe(p, plan, R):
    p.update_partials(plan.ops)
    e = plan.root_edge
    lnl, ps = p.compute_edge_loglikelihood(*e, [0] * R, persite=True)
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    d = [p.compute_likelihood_derivatives(e[1], e[3], t, [0] * R, st) for t in (0.03, 0.4)]
    p.update_prob_matrices([0] * R, [int(plan.ops[0]["child1_matrix_index"])], [0.2])
    p.update_partials(plan.ops)
    lnl2 = p.compute_edge_loglikelihood(*e, [0] * R)
    return lnl, ps, np.array(d), lnl2


for seed in range(first, first + count):
    if budget_s and time.time() - t0 > budget_s:
        break
    done += 1
    rng = np.random.default_rng(50_000 + seed)
    states = 4 if rng.random() < 0.7 else 20
    shape = ("random", "balanced", "caterpillar")[seed % 3]
    tips = int(2 ** rng.integers(2, 6)) if shape == "balanced" else int(rng.integers(4, 60))
    sites = int(rng.integers(300, 6000))
    k = int(rng.integers(2, 9))
    if fixed_k:
        k = fixed_k
        sites = max(sites, 256 * k + 300)   # (eight non-empty ranges on multiples of 256 sites)
    attrs = ATTRIB_PATTERN_TIP | (ATTRIB_RATE_SCALERS if rng.random() < 0.3 else 0)
    rep = ATTRIB_SITE_REPEATS if rng.random() < 0.25 else 0
    case = make_case(states, shape, tips, sites, seed=seed)
    plan, R = case["plan"], case["rate_cats"]
    whole = build_partition(amd, case, attrs)
    want = observe(whole, plan, R)
    whole.destroy()
    devs = (C.c_int * k)(*([0] * k))
    assert amd.lib.pll_amd_set_devices(devs, k) == 1
    p = build_partition(amd, case, attrs | rep)
    amd.lib.pll_amd_set_devices(None, 0)
    got = observe(p, plan, R)
    if seed % 7 == 0:   # an error on the shards' threads, then on with it
        bad_ops = plan.ops.copy()
        bad_ops["child1_clv_index"][0] = 10 ** 6
        amd.clear_error()
        p.update_partials(bad_ops)
        ok_err = amd.errno() != 0
        p.update_partials(plan.ops)
        again = p.compute_edge_loglikelihood(*plan.root_edge, [0] * R)
        ok_err = ok_err and again == got[3]
    else:
        ok_err = True
    p.destroy()
    ok = bits_equal(got[1], want[1]) and abs(got[0] - want[0]) <= 1e-12 * abs(want[0]) and \
        abs(got[3] - want[3]) <= 1e-12 * abs(want[3]) and rel_err(got[2], want[2]) < 1e-11 and ok_err
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, states, shape, tips, sites, k, attrs, rep, flush=True)
print("soak_shards: %d seeds from %d%s, %d mismatches, %.0f s on %d core(s)" % (
    done, first, ", %d shards each" % fixed_k if fixed_k else "", bad, time.time() - t0, len(os.sched_getaffinity(0))))
sys.exit(1 if bad else 0)
