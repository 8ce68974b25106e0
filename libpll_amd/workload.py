"""Synthetic phylogenetic workloads: tree shapes as pll_operation_t lists,
alignments, models.  Host-side numpy only; nothing here touches the GPU.

The reference builds op lists with pll_utree_traverse + pll_utree_create_operations
(utree.c:403,284; out of scope here) or by hand in its tests
(test/src/00010_NMDU_lkcalc.c:151-175).  We build them by hand the same way:

  node numbering   tips 0..T-1, inner nodes T..2T-3 (one CLV buffer each)
  scalers          inner node v uses scale buffer v-T
  P-matrices       the edge above node v uses matrix index v
  ops              every inner node once, children before parents
  root edge        the single edge joining the last two subtrees (unrooted tree)

SURVEY.md section 8(d) fixes the generator: splitmix64 stream, seed 42, branch
lengths U(0.01, 0.2), GTR rates / frequencies below, Gamma alpha 0.7 with 4
mean-discretised categories, 2 % fully ambiguous characters.
"""
from dataclasses import dataclass, field

import numpy as np

from .pllapi import OPS_DTYPE, SCALE_BUFFER_NONE

GTR_RATES = np.array([1.2, 3.1, 0.9, 1.1, 3.4, 1.0])
GTR_FREQS = np.array([0.28, 0.22, 0.24, 0.26])
GAMMA_ALPHA = 0.7
DNA_CHARS = b"ACGT"
AA_CHARS = b"ARNDCQEGHILKMFPSTWYV"


class SplitMix64:
    """The splitmix64 generator (Steele, Lea & Flood 2014): tiny, seedable and
    identical in every language a caller may reimplement it in."""

    def __init__(self, seed):
        self.x = np.uint64(seed)

    def next_u64(self):
        with np.errstate(over="ignore"):
            self.x = self.x + np.uint64(0x9E3779B97F4A7C15)
            z = self.x
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * (int(self.next_u64() >> np.uint64(11)) / float(1 << 53))

    def below(self, n):
        return int(self.next_u64() % np.uint64(n))


@dataclass
class TreePlan:
    tips: int
    ops: np.ndarray                 # OPS_DTYPE, post-order
    matrix_indices: np.ndarray      # every edge's matrix index
    branch_lengths: np.ndarray      # same length
    root_edge: tuple                # (parent_clv, parent_scaler, child_clv, child_scaler, matrix)
    parent_of: dict = field(default_factory=dict)   # node -> (parent node), for simulation
    shape: str = ""

    @property
    def clv_buffers(self):
        return self.tips - 2

    @property
    def scale_buffers(self):
        return self.tips - 2

    @property
    def prob_matrices(self):
        return 2 * self.tips - 2

    def op_kinds(self):
        """(tip-tip, tip-inner, inner-inner) op counts under PATTERN_TIP."""
        t = self.tips
        c1 = self.ops["child1_clv_index"] < t
        c2 = self.ops["child2_clv_index"] < t
        return int((c1 & c2).sum()), int((c1 ^ c2).sum()), int((~c1 & ~c2).sum())


def _scaler_of(node, tips, use_scalers):
    return node - tips if (use_scalers and node >= tips) else SCALE_BUFFER_NONE


def _assemble(tips, joins, last_two, rng, shape, use_scalers=True, branch=None):
    ops = np.zeros(len(joins), dtype=OPS_DTYPE)
    parent_of = {}
    for i, (parent, a, b) in enumerate(joins):
        ops[i] = (parent, _scaler_of(parent, tips, use_scalers),
                  a, a, _scaler_of(a, tips, use_scalers),
                  b, b, _scaler_of(b, tips, use_scalers))
        parent_of[a] = parent
        parent_of[b] = parent
    u, v = last_two
    # the root edge uses u's matrix slot; v's slot stays unused
    parent_of[v] = u
    edges = sorted(set(parent_of.keys()) - {v}) + [u]
    edges = sorted(set(edges))
    mi = np.array(edges, dtype=np.uint32)
    if branch is None:
        bl = np.array([rng.uniform(0.01, 0.2) for _ in edges])
    else:
        bl = np.full(len(edges), float(branch))
    root = (u, _scaler_of(u, tips, use_scalers), v, _scaler_of(v, tips, use_scalers), u)
    return TreePlan(tips, ops, mi, bl, root, parent_of, shape)


def balanced_tree(tips, seed=42, use_scalers=True, branch=None):
    """Perfectly balanced binary tree over `tips` (a power of two >= 4),
    unrooted at the top split: T/2 tip-tip ops, then T/2-2 inner-inner ops."""
    assert tips >= 4 and tips & (tips - 1) == 0
    rng = SplitMix64(seed)
    level = list(range(tips))
    nxt = tips
    joins = []
    while len(level) > 2:
        new = []
        for i in range(0, len(level), 2):
            joins.append((nxt, level[i], level[i + 1]))
            new.append(nxt)
            nxt += 1
        level = new
    return _assemble(tips, joins, (level[0], level[1]), rng, "balanced", use_scalers, branch)


def caterpillar_tree(tips, seed=42, use_scalers=True, branch=None):
    """Ladder: ((((t0,t1),t2),t3)...); the root edge joins the last inner node
    and the last tip.  Depth = tips-2, which drives scaler counts up."""
    assert tips >= 3
    rng = SplitMix64(seed)
    joins = [(tips, 0, 1)]
    for k in range(1, tips - 2):
        joins.append((tips + k, tips + k - 1, k + 1))
    return _assemble(tips, joins, (2 * tips - 3, tips - 1), rng, "caterpillar", use_scalers, branch)


def random_tree(tips, seed=42, use_scalers=True, branch=None):
    """Random joining order (Yule-like topology)."""
    assert tips >= 3
    rng = SplitMix64(seed)
    active = list(range(tips))
    nxt = tips
    joins = []
    while len(active) > 2:
        i = rng.below(len(active))
        a = active.pop(i)
        j = rng.below(len(active))
        b = active.pop(j)
        joins.append((nxt, a, b))
        active.append(nxt)
        nxt += 1
    u, v = active
    if u < tips and v >= tips:
        u, v = v, u
    return _assemble(tips, joins, (u, v), rng, "random", use_scalers, branch)


# ---- op lists that change: what a tree search hands pll_update_partials ------------------------
class UnrootedView:
    """The tree of a TreePlan as an unrooted tree: every inner node has one CLV / scale buffer (its
    node number, as in TreePlan) that holds the node's partial towards WHICHEVER neighbour the last
    traversal was directed at -- the way libpll's unrooted trees share a clv_index among the three
    directions of a node (pll.h:312-324; test/src/partial-traversal.c:17-58 re-orients them)."""

    def __init__(self, plan, use_scalers=True):
        self.tips = plan.tips
        self.use_scalers = use_scalers
        self.adj = {}
        self.matrix = {}
        u, _, v, _, m = plan.root_edge
        for child, parent in plan.parent_of.items():
            if child == v and parent == u:
                continue
            self._edge(child, parent, child)      # an edge's matrix slot is its child's number
        self._edge(u, v, m)
        self.root = (u, v)

    def _edge(self, a, b, m):
        self.adj.setdefault(a, []).append(b)
        self.adj.setdefault(b, []).append(a)
        self.matrix[frozenset((a, b))] = m

    def edges(self):
        return [tuple(sorted(e)) for e in self.matrix]

    def traversal(self, root):
        """Post-order op list of the whole tree directed at the edge `root` = (a, b), and the
        arguments of pll_compute_edge_loglikelihood at that edge."""
        a, b = root
        ops = []
        for top, away in ((a, b), (b, a)):
            stack = [(top, away, False)]
            while stack:
                x, frm, done = stack.pop()
                if x < self.tips:
                    continue
                kids = [y for y in self.adj[x] if y != frm]
                if not done:
                    stack.append((x, frm, True))
                    for y in kids:
                        stack.append((y, x, False))
                else:
                    c1, c2 = kids
                    ops.append((x, _scaler_of(x, self.tips, self.use_scalers),
                                c1, self.matrix[frozenset((x, c1))], _scaler_of(c1, self.tips, self.use_scalers),
                                c2, self.matrix[frozenset((x, c2))], _scaler_of(c2, self.tips, self.use_scalers)))
        arr = np.zeros(len(ops), dtype=OPS_DTYPE)
        for i, o in enumerate(ops):
            arr[i] = o
        # (a tip may only be the "child" side of the call: the inner node first)
        if a < self.tips:
            a, b = b, a
        edge = (a, _scaler_of(a, self.tips, self.use_scalers), b, _scaler_of(b, self.tips, self.use_scalers),
                self.matrix[frozenset((a, b))])
        return arr, edge

    def partial(self, full_ops, changed_edges, root):
        """The ops of `full_ops` (a traversal directed at `root`) that a change of the branches
        `changed_edges` invalidates: the nodes between each branch and the root edge."""
        parent = {}
        for op in full_ops:
            p = int(op["parent_clv_index"])
            parent[int(op["child1_clv_index"])] = p
            parent[int(op["child2_clv_index"])] = p
        dirty = set()
        for x, y in changed_edges:
            # (the branch hangs below y if y is x's parent in this orientation, below x the other way
            # round; the root edge itself invalidates no CLV)
            node = y if parent.get(x) == y else x if parent.get(y) == x else None
            while node is not None:
                dirty.add(node)
                node = parent.get(node)
        keep = np.array([int(op["parent_clv_index"]) in dirty for op in full_ops], dtype=bool)
        return full_ops[keep]


# ---- models ---------------------------------------------------------------------

def q_matrix(rates, freqs):
    """Reversible rate matrix normalised to one substitution per unit time."""
    n = len(freqs)
    q = np.zeros((n, n))
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            q[i, j] = rates[k] * freqs[j]
            q[j, i] = rates[k] * freqs[i]
            k += 1
    q -= np.diag(q.sum(axis=1))
    return q / -(freqs * np.diag(q)).sum()


def transition_matrix(q, t):
    w, v = np.linalg.eig(q)
    p = (v * np.exp(w * t)) @ np.linalg.inv(v)
    p = np.clip(p.real, 0.0, None)
    return p / p.sum(axis=1, keepdims=True)


# ---- alignments -------------------------------------------------------------------

def random_alignment(tips, sites, states=4, seed=42, gap_frac=0.02):
    """i.i.d. uniform tip characters with `gap_frac` fully ambiguous ('-').
    Returns a list of `tips` bytes objects of length `sites`."""
    chars = np.frombuffer(DNA_CHARS if states == 4 else AA_CHARS, dtype=np.uint8)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(tips):
        s = chars[rng.integers(0, len(chars), size=sites)]
        if gap_frac > 0:
            s = np.where(rng.random(sites) < gap_frac, np.uint8(ord("-")), s)
        out.append(s.astype(np.uint8).tobytes())
    return out


def simulated_alignment(plan, sites, rates, freqs, cat_rates, seed=42, gap_frac=0.02):
    """Evolve `sites` characters down the tree under the model (root state from
    the stationary distribution at node root_edge[0], one Gamma category per
    site), so that CLVs shrink the way they do on real data."""
    states = len(freqs)
    chars = np.frombuffer(DNA_CHARS if states == 4 else AA_CHARS, dtype=np.uint8)
    rng = np.random.default_rng(seed)
    q = q_matrix(rates, freqs)
    cat = rng.integers(0, len(cat_rates), size=sites)
    blen = dict(zip(plan.matrix_indices.tolist(), plan.branch_lengths.tolist()))
    u, _, v, _, m = plan.root_edge
    state = {u: rng.choice(states, size=sites, p=freqs / freqs.sum())}
    children = {}
    for node, par in plan.parent_of.items():
        children.setdefault(par, []).append(node)
    stack = [u]
    while stack:
        par = stack.pop()
        for ch in children.get(par, []):
            t = blen[m] if (par == u and ch == v) else blen[ch]
            new = np.empty(sites, dtype=np.int64)
            for k, r in enumerate(cat_rates):
                sel = np.nonzero(cat == k)[0]
                if not len(sel):
                    continue
                cum = np.cumsum(transition_matrix(q, t * r), axis=1)
                draw = rng.random(len(sel))
                new[sel] = (draw[:, None] > cum[state[par][sel]]).sum(axis=1).clip(0, states - 1)
            state[ch] = new
            stack.append(ch)
        if par >= plan.tips:
            del state[par]
    out = []
    for tip in range(plan.tips):
        s = chars[state[tip]]
        if gap_frac > 0:
            s = np.where(rng.random(sites) < gap_frac, np.uint8(ord("-")), s)
        out.append(s.astype(np.uint8).tobytes())
    return out


ALIGNMENT_BLOCK = 250_000


def global_alignment(plan, lo, hi, rates, freqs, cat_rates, seed=42, block=ALIGNMENT_BLOCK, distinct=0,
                     cache=None, kind="simulated"):
    """Columns [lo, hi) of ONE alignment that is defined block by block, so that every rank of a
    multi-GPU job can make its own site range without anybody making the whole: block b -- sites
    [b * block, (b + 1) * block) -- is `simulated_alignment` with seed + b (distinct = 0) or, to bound
    the cost of a very long alignment, seed + b % distinct (the alignment then repeats after
    `distinct` blocks; kind = "random": i.i.d. characters instead).  A range is a slice of the whole by construction: rank r's [lo, hi) of an
    N-GPU run are the very columns a one-GPU run evaluates at positions lo..hi-1, which is what makes
    the N-GPU lnL checkable (the reference's sum runs over ONE alignment,
    core_likelihood_avx.c:1246-1259).  cache: a dict that keeps generated blocks between calls."""
    if not 0 <= lo <= hi:
        raise ValueError("bad site range [%d, %d)" % (lo, hi))
    cache = {} if cache is None else cache
    parts = [[] for _ in range(plan.tips)]
    for b in range(lo // block, -(-hi // block) if hi > lo else lo // block):
        key = b % distinct if distinct else b
        if key not in cache:
            cache[key] = (simulated_alignment(plan, block, rates, freqs, cat_rates, seed=seed + key) if kind == "simulated"
                          else random_alignment(plan.tips, block, len(freqs), seed=seed + key))
        first, last = max(lo, b * block) - b * block, min(hi, (b + 1) * block) - b * block
        for t in range(plan.tips):
            parts[t].append(cache[key][t][first:last])
    return [b"".join(x) for x in parts]


def reference_lnl(ref, plan, seqs, states, rate_cats, attributes, chunk=50_000, budget_s=None, root_edge=None):
    """lnL of an alignment through a CPU library with the reference's API (oracle/_ref in bench.py and
    the tests), in site chunks that reuse ONE small partition: lnL is a sum over sites, and a partition
    of the whole would need the alignment's CLVs in host memory (133 GB for BASELINE config 4).
    Returns (lnl, sites evaluated): all of them, or -- with budget_s -- as many whole chunks as fit the
    time budget (at least one), so that a caller can bound the check and compare the same prefix."""
    import time
    sites = len(seqs[0])
    edge = plan.root_edge if root_edge is None else root_edge
    cmap = ref.map("nt" if states == 4 else "aa")
    total, done, part, t0 = 0.0, 0, None, time.perf_counter()
    while done < sites:
        n = min(chunk, sites - done)
        piece = [s[done:done + n] for s in seqs]
        if part is None or n != chunk:
            if part is not None:
                part.destroy()
            part = setup_partition(ref, plan, piece, states, rate_cats, attributes)
        else:
            for i, s in enumerate(piece):
                part.set_tip_states(i, cmap, s)
        part.update_partials(plan.ops)
        total += part.compute_edge_loglikelihood(*edge, [0] * rate_cats)
        done += n
        if budget_s is not None and time.perf_counter() - t0 > budget_s:
            break
    if part is not None:
        part.destroy()
    return total, done


# ---- site sharding (multi-GPU) -------------------------------------------------------

def shard_bounds(sites, nranks, granule=256):
    """Contiguous site ranges for `nranks` devices, boundaries on multiples of
    `granule` sites (keeps every per-site array 16-byte aligned).  Returns
    nranks+1 offsets."""
    per = -(-sites // nranks)
    per = -(-per // granule) * granule
    bounds = [min(sites, r * per) for r in range(nranks)] + [sites]
    if any(bounds[r + 1] <= bounds[r] for r in range(nranks)):
        # every rank computes the same bounds, so every rank raises here -- before any
        # collective a rank with an empty range would leave the others waiting in
        raise ValueError("%d sites cannot be split into %d non-empty ranges on multiples of %d sites"
                         % (sites, nranks, granule))
    return bounds


# ---- one-call setup used by tests and bench.py -----------------------------------------

def setup_partition(lib, plan, seqs, states=4, rate_cats=4, attributes=0, alpha=GAMMA_ALPHA,
                    rates=None, freqs=None, pattern_weights=None, pinv=0.0, site_range=None):
    """Create a partition on `lib` (product or reference), load model and tips,
    compute all P-matrices.  `site_range` = (lo, hi) keeps only those alignment
    columns (site sharding)."""
    if site_range is not None:
        lo, hi = site_range
        seqs = [s[lo:hi] for s in seqs]
        if pattern_weights is not None:
            pattern_weights = pattern_weights[lo:hi]
    sites = len(seqs[0])
    if rates is None:
        rates = GTR_RATES if states == 4 else lib.aa_model("lg")[0]
    if freqs is None:
        freqs = GTR_FREQS if states == 4 else lib.aa_model("lg")[1]
    p = lib.partition_create(plan.tips, plan.clv_buffers, states, sites, 1, plan.prob_matrices,
                             rate_cats, plan.scale_buffers, attributes)
    p.set_frequencies(0, freqs)
    p.set_subst_params(0, rates)
    p.set_category_rates(lib.compute_gamma_cats(alpha, rate_cats))
    cmap = lib.map("nt" if states == 4 else "aa")
    for i, s in enumerate(seqs):
        p.set_tip_states(i, cmap, s)
    if pattern_weights is not None:
        p.set_pattern_weights(pattern_weights)
    if pinv > 0:
        p.update_invariant_sites_proportion(0, pinv)
    p.update_prob_matrices([0] * rate_cats, plan.matrix_indices, plan.branch_lengths)
    return p
