"""Shared builders for the parity tests: one description of a case, applied
identically to a pll-API library (product or reference) and to the oracle."""
import numpy as np

from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, OPS_DTYPE, SCALE_BUFFER_NONE
from oracle_api import OracleRun

TREES = {"balanced": W.balanced_tree, "caterpillar": W.caterpillar_tree, "random": W.random_tree}


def make_case(states=4, shape="balanced", tips=16, sites=200, rate_cats=4, seed=1, alpha=0.7,
              gap_frac=0.05, weights=True, branch=None, ambiguity=True):
    """A tree + alignment + model description (plain data)."""
    plan = TREES[shape](tips, seed=seed, branch=branch)
    rng = np.random.default_rng(seed)
    seqs = None
    if states in (4, 20):
        seqs = W.random_alignment(tips, sites, states, seed=seed + 100, gap_frac=gap_frac)
        if ambiguity:
            extra = b"RYKMSWBDHVN" if states == 4 else b"BZX"
            seqs = [bytearray(s) for s in seqs]
            for s in seqs:
                for pos in rng.integers(0, sites, size=max(1, sites // 25)):
                    s[pos] = extra[rng.integers(0, len(extra))]
            seqs = [bytes(s) for s in seqs]
    pw = rng.integers(1, 4, size=sites).astype(np.uint32) if weights else None
    if states == 4:
        rates, freqs = W.GTR_RATES, W.GTR_FREQS
    else:
        rates = rng.uniform(0.3, 4.0, states * (states - 1) // 2)
        freqs = rng.dirichlet(np.ones(states) * 8)
    return dict(states=states, plan=plan, seqs=seqs, sites=sites, rate_cats=rate_cats,
                alpha=alpha, pw=pw, rates=rates, freqs=freqs, seed=seed, tips=tips, cmap=None)


ALPHABET = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"


def odd_state_case(states, tips=9, sites=30, seed=3, shape="random", **kw):
    """Data with 2..32 states (other than 4 and 20) over the alphabet A.. with a
    hand-made character map."""
    case = make_case(states, shape, tips, sites, seed=seed, **kw)
    rng = np.random.default_rng(seed)
    alphabet = ALPHABET[:states]
    chars = np.frombuffer(alphabet, dtype=np.uint8)
    case["seqs"] = [chars[rng.integers(0, states, sites)].tobytes() for _ in range(tips)]
    cmap = np.zeros(256, dtype=np.uint32)
    for i, ch in enumerate(alphabet):
        cmap[ch] = 1 << i
    cmap[ord("-")] = (1 << states) - 1
    for s in range(0, tips, 3):
        b = bytearray(case["seqs"][s])
        b[s % sites] = ord("-")
        case["seqs"][s] = bytes(b)
    case["cmap"] = cmap
    return case


def many_state_case(states, tips=9, sites=30, seed=3, shape="random", **kw):
    """More than 32 states (61 = codons): the character maps of the API are 32-bit masks,
    so such data enters as tip CLVs; case["tip_index"][tip][site] = state, -1 = gap."""
    case = make_case(states, shape, tips, sites, seed=seed, **kw)
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, states, size=(tips, sites))
    idx[rng.random((tips, sites)) < 0.03] = -1
    case["tip_index"] = idx
    return case


def index_tip_clvs(case):
    S, R = case["states"], case["rate_cats"]
    idx = case["tip_index"]
    out = (idx[:, :, None] == np.arange(S)[None, None, :]) | (idx[:, :, None] < 0)
    return np.repeat(out[:, :, None, :].astype(np.float64), R, axis=2)


def case_map(lib, case):
    if case["cmap"] is not None:
        return case["cmap"]
    return lib.map("nt" if case["states"] == 4 else "aa")


def build_partition(lib, case, attrs, pinv=0.0):
    """Create + fill a partition on `lib`; returns it with P-matrices computed."""
    plan, S, R = case["plan"], case["states"], case["rate_cats"]
    if lib.is_amd:
        attrs &= ~0xF
    p = lib.partition_create(plan.tips, plan.clv_buffers, S, case["sites"], 1, plan.prob_matrices,
                             R, plan.scale_buffers, attrs)
    p.set_frequencies(0, case["freqs"])
    p.set_subst_params(0, case["rates"])
    p.set_category_rates(lib.compute_gamma_cats(case["alpha"], R))
    if case.get("tip_index") is not None:
        for i, clv in enumerate(index_tip_clvs(case)):
            p.set_tip_clv(i, clv[:, 0, :].reshape(-1))   # [sites][states]; the call replicates over rates
    else:
        cmap = case_map(lib, case)
        for i, s in enumerate(case["seqs"]):
            p.set_tip_states(i, cmap, s)
    if case["pw"] is not None:
        p.set_pattern_weights(case["pw"])
    if pinv > 0:
        p.update_invariant_sites_proportion(0, pinv)
    p.update_prob_matrices([0] * R, plan.matrix_indices, plan.branch_lengths)
    return p


def model_of(part, lib, case, pinv=0.0):
    """Model arrays as the partition holds them on the host (eigen system from
    pll_update_eigen, category rates from pll_compute_gamma_cats)."""
    vals, vecs, inv = part.get_eigen(0)
    S, R = case["states"], case["rate_cats"]
    fr = np.ctypeslib.as_array(part.s.frequencies[0], shape=(part.s.states_padded,)).copy()[:S]
    return dict(states=S, rate_cats=R, rates=lib.compute_gamma_cats(case["alpha"], R),
                rate_weights=np.full(R, 1.0 / R), eigenvals=vals, eigenvecs=vecs,
                inv_eigenvecs=inv, freqs=fr, pinv=pinv)


def encode_tips(part):
    """(tipcodes[tips][sites] uint8, tipmap) as the partition encoded them."""
    s = part.s
    codes = np.stack([np.ctypeslib.as_array(s.tipchars[i], shape=(s.sites,)).copy()
                      for i in range(s.tips)])
    tipmap = np.ctypeslib.as_array(s.tipmap, shape=(256,)).copy()
    return codes, tipmap


def tip_clvs(case, cmap):
    """0/1 tip CLVs [tips][sites][R][S] from the character map (pll.c:905-939)."""
    S, R = case["states"], case["rate_cats"]
    out = np.zeros((case["tips"], case["sites"], R, S))
    for i, seq in enumerate(case["seqs"]):
        masks = cmap[np.frombuffer(seq, dtype=np.uint8)]
        bits = (masks[:, None] >> np.arange(S)[None, :]) & 1
        out[i] = bits[:, None, :].astype(np.float64)
    return out


def invariant_of(part):
    s = part.s
    if not s.invariant:
        return None
    return np.ctypeslib.as_array(s.invariant, shape=(s.sites,)).copy()


def oracle_run(orc, lib, part, case, attrs, pinv=0.0):
    model = model_of(part, lib, case, pinv)
    inv = invariant_of(part) if pinv > 0 else None
    if attrs & ATTRIB_PATTERN_TIP:
        codes, tipmap = encode_tips(part)
        return OracleRun(orc, model, case["plan"], attrs, tipcodes=codes, tipmap=tipmap,
                         pattern_weights=case["pw"], invariant=inv)
    if case.get("tip_index") is not None:
        return OracleRun(orc, model, case["plan"], attrs, tipclvs=index_tip_clvs(case),
                         pattern_weights=case["pw"], invariant=inv)
    return OracleRun(orc, model, case["plan"], attrs, tipclvs=tip_clvs(case, case_map(lib, case)),
                     pattern_weights=case["pw"], invariant=inv)


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and bool((a.view(np.uint64) == b.view(np.uint64)).all())


def clvs_bitwise(states):
    """Does the configuration the environment selects promise every CLV bit for bit?  Not 20 states on the default
    path: PLLHIP_AA_TI_MFMA (default on) runs tip-inner mat-vecs of the whole-list kernel on the matrix cores."""
    import os
    return not (states == 20 and os.environ.get("PLLHIP_AA_TI_MFMA", "1") != "0" and
                os.environ.get("PLLHIP_AA_EXACT", "0") != "1")


def clv_err(a, b):
    """Largest difference of two CLVs [sites][rates][states], entry by entry, relative to the entry -- or to 1e-150 x
    the site's largest entry, if that is more.  The floor is for entries that were DENORMAL on the way (a block whose
    largest entry is kept above 2^-256 by the scaling can hold entries 200 orders of magnitude below it): they were
    rounded to a multiple of 2^-1074 there, so two summation orders that differ in the last bit of a normal number
    differ by 1e-12 relative in such an entry (seen: 6.4e-235 in a block with 2.4e-67 -- exactly 2^-1074 x 2^256
    apart).  Such entries weigh nothing in any likelihood."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not a.size:
        return 0.0
    floor = 1e-150 * np.abs(b).reshape(b.shape[0], -1).max(axis=1).reshape((-1,) + (1,) * (b.ndim - 1))
    return float(np.max(np.abs(a - b) / np.maximum(np.maximum(np.abs(b), floor), 1e-300)))


def clv_ok(a, b, exact=True, tol=1e-12):
    """A CLV against the oracle's / the reference's: bit for bit, or -- 20 states on the default path, where the
    whole-list kernel runs the mat-vec of tip-inner ops on the matrix cores (round 6) -- to rounding, entry by entry
    (clv_err; ~6e-16 per op of depth measured on a 300-tip ladder: 1e-13 for the trees of the BASELINE configs, which
    tests/test_gpu_baseline_configs.py asks for, 1e-12 by default -- 400-tip ladders).
    (Scaler counts are compared bit for bit either way: the scaling certificate, tests/test_gpu_cert.py.)"""
    return bits_equal(a, b) if exact else clv_err(a, b) <= tol


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def sumtable_err(a, b):
    """max |a-b| relative to the largest entry of the same (site, rate) block:
    entries that are analytically zero (gap columns) carry only rounding noise."""
    scale = np.abs(b).max(axis=2, keepdims=True) + 1e-300
    return float(np.max(np.abs(a - b) / scale))


def random_op_sequence(rng, tips, inner, scalers, matrices, length):
    """A valid-but-arbitrary op sequence over `inner` CLV slots: every op reads
    tips or slots written earlier and writes any slot other than its children.
    Produces every kind of dependency between neighbouring ops (read-after-write,
    write-after-read, write-after-write, shared scaler slots)."""
    ops = np.zeros(length, dtype=OPS_DTYPE)
    written = {}                      # inner CLV slot -> scaler slot that goes with it (or NONE)
    recent = []
    for i in range(length):
        pool = list(range(tips)) + sorted(written)

        def child():
            # mostly recent results, so that deep chains (and scaling events) build up
            if recent and rng.random() < 0.7:
                c = int(recent[rng.integers(0, len(recent))])
            else:
                c = int(pool[rng.integers(0, len(pool))])
            return c, (written[c] if c >= tips else SCALE_BUFFER_NONE)
        (a, sa), (b, sb) = child(), child()
        free = [s for s in range(tips, tips + inner) if s not in (a, b)]
        # bias towards a few slots so that reuse is frequent
        parent = int(free[min(int(rng.exponential(3.0)), len(free) - 1)])
        psc = int(rng.integers(0, scalers)) if rng.random() < 0.8 else SCALE_BUFFER_NONE
        # the reference adds child counts into the parent's buffer: a parent sharing its
        # scaler slot with one of its children would read what it is overwriting
        if psc in (sa, sb):
            psc = SCALE_BUFFER_NONE
        ops[i] = (parent, psc, a, int(rng.integers(0, matrices)), sa, b, int(rng.integers(0, matrices)), sb)
        # any other CLV that used this scaler slot loses it
        for k in list(written):
            if psc != SCALE_BUFFER_NONE and written[k] == psc:
                written[k] = SCALE_BUFFER_NONE
        written[parent] = psc
        recent = ([parent] + [r for r in recent if r != parent])[:3]
    return ops


def random_sequence_case(seed, states=None, rate_cats=4):
    """(case, attributes, ops, rng) for the random-op-sequence tests: 4- and 20-state
    data (or the state count given), with and without PATTERN_TIP, per-site and per-rate
    scalers."""
    rng = np.random.default_rng(1000 + seed)
    attrs = (ATTRIB_PATTERN_TIP if seed % 2 else 0) | (ATTRIB_RATE_SCALERS if seed % 4 >= 2 else 0)
    tips = 12
    sites = 97 + 64 * (seed % 4)
    if states is None:
        states = 4 if seed % 3 else 20
        case = make_case(states, "random", tips, sites, seed=seed + 50)
    elif states > 32:
        attrs &= ~ATTRIB_PATTERN_TIP
        case = many_state_case(states, tips=tips, sites=sites, seed=seed + 50, rate_cats=rate_cats)
    else:
        case = odd_state_case(states, tips=tips, sites=sites, seed=seed + 50, rate_cats=rate_cats)
    plan = case["plan"]
    ops = random_op_sequence(rng, tips, plan.clv_buffers, plan.scale_buffers, plan.prob_matrices - 1,
                             120)
    # every matrix slot gets its own branch length
    plan.matrix_indices = np.arange(plan.prob_matrices - 1, dtype=np.uint32)
    plan.branch_lengths = rng.uniform(0.01, 0.5, len(plan.matrix_indices))
    return case, attrs, ops, rng
