"""CPU-only parity of the two host-side steps either side of the hot path
(SURVEY 8f): tree -> op list (pll_utree/rtree_traverse + _create_operations) and
alignment -> unique site patterns (pll_compress_site_patterns), product vs the
genuine reference on the same in-memory inputs.  Exact equality everywhere."""
import ctypes as C

import numpy as np
import pytest

from libpll_amd.pllapi import Operation


class UNode(C.Structure):
    pass


UNode._fields_ = [("label", C.c_char_p), ("length", C.c_double), ("node_index", C.c_uint),
                  ("clv_index", C.c_uint), ("scaler_index", C.c_int), ("pmatrix_index", C.c_uint),
                  ("next", C.POINTER(UNode)), ("back", C.POINTER(UNode)), ("data", C.c_void_p)]


class RNode(C.Structure):
    pass


RNode._fields_ = [("label", C.c_char_p), ("length", C.c_double), ("node_index", C.c_uint),
                  ("clv_index", C.c_uint), ("scaler_index", C.c_int), ("pmatrix_index", C.c_uint),
                  ("left", C.POINTER(RNode)), ("right", C.POINTER(RNode)),
                  ("parent", C.POINTER(RNode)), ("data", C.c_void_p)]


def random_utree(tips, rng):
    """Random unrooted binary tree as the reference's parser would lay it out:
    tips are single nodes, inner nodes rings of three; returns (all node structs,
    a root handle that is an inner node)."""
    nodes = []

    def new(clv, scaler):
        n = UNode()
        n.clv_index, n.scaler_index = clv, scaler
        n.length = float(rng.uniform(0.01, 0.5))
        nodes.append(n)
        return n

    def link(a, b, pm):
        a.back = C.pointer(b)
        b.back = C.pointer(a)
        b.length = a.length
        a.pmatrix_index = b.pmatrix_index = pm

    # dangling half-edges of the growing forest
    open_ends = [new(i, -1) for i in range(tips)]
    inner = 0
    pm = 0
    while len(open_ends) > 3:
        i, j = sorted(rng.choice(len(open_ends), 2, replace=False), reverse=True)
        a, b = open_ends.pop(i), open_ends.pop(j)
        ring = [new(tips + inner, inner) for _ in range(3)]
        for k in range(3):
            ring[k].next = C.pointer(ring[(k + 1) % 3])
        link(a, ring[1], pm)
        link(b, ring[2], pm + 1)
        pm += 2
        inner += 1
        open_ends.append(ring[0])
    ring = [new(tips + inner, inner) for _ in range(3)]
    for k in range(3):
        ring[k].next = C.pointer(ring[(k + 1) % 3])
        link(open_ends[k], ring[k], pm + k)
    return nodes, ring[0]


def run_utree(lib, root, trav, n_nodes, pruned=()):
    pruned = set(pruned)
    CB = C.CFUNCTYPE(C.c_int, C.POINTER(UNode))
    cb = CB(lambda p: 0 if p.contents.clv_index in pruned else 1)
    buf = (C.POINTER(UNode) * n_nodes)()
    size = C.c_uint()
    lib.pll_utree_traverse.argtypes = [C.POINTER(UNode), C.c_int, CB, C.POINTER(C.POINTER(UNode)),
                                       C.POINTER(C.c_uint)]
    rc = lib.pll_utree_traverse(C.pointer(root), trav, cb, buf, C.byref(size))
    order = [(buf[i].contents.clv_index, buf[i].contents.pmatrix_index) for i in range(size.value)]
    branches = (C.c_double * n_nodes)()
    pmi = (C.c_uint * n_nodes)()
    ops = (Operation * n_nodes)()
    mc, oc = C.c_uint(), C.c_uint()
    lib.pll_utree_create_operations.restype = None
    lib.pll_utree_create_operations.argtypes = [C.POINTER(C.POINTER(UNode)), C.c_uint,
                                                C.POINTER(C.c_double), C.POINTER(C.c_uint),
                                                C.POINTER(Operation), C.POINTER(C.c_uint),
                                                C.POINTER(C.c_uint)]
    lib.pll_utree_create_operations(buf, size, branches, pmi, ops, C.byref(mc), C.byref(oc))
    ops_t = [tuple(getattr(ops[i], f) for f, _ in Operation._fields_) for i in range(oc.value)]
    return rc, order, list(branches[:mc.value]), list(pmi[:mc.value]), ops_t


@pytest.mark.parametrize("tips", [3, 4, 5, 17, 200])
def test_utree_traverse_and_operations(amd, ref_tree, tips):
    ref = ref_tree
    rng = np.random.default_rng(tips)
    nodes, root = random_utree(tips, rng)
    for trav in (1, 2):   # post-order, pre-order
        for pruned in ((), (tips + 1, 2)):
            a = run_utree(amd.lib, root, trav, len(nodes), pruned)
            r = run_utree(ref.lib, root, trav, len(nodes), pruned)
            assert a == r
    full = run_utree(amd.lib, root, 1, len(nodes))
    assert len(full[4]) == tips - 2 and len(full[2]) == 2 * tips - 3
    # a tip as root is refused, as in the reference (utree.c:410)
    tip = next(n for n in nodes if not n.next)
    assert run_utree(amd.lib, tip, 1, len(nodes))[0] == run_utree(ref.lib, tip, 1, len(nodes))[0] == 0


def random_rtree(tips, rng):
    nodes = []

    def new(clv, scaler):
        n = RNode()
        n.clv_index, n.scaler_index, n.pmatrix_index = clv, scaler, clv
        n.length = float(rng.uniform(0.01, 0.5))
        nodes.append(n)
        return n

    active = [new(i, -1) for i in range(tips)]
    inner = 0
    while len(active) > 1:
        i, j = sorted(rng.choice(len(active), 2, replace=False), reverse=True)
        a, b = active.pop(i), active.pop(j)
        p = new(tips + inner, inner)
        p.left, p.right = C.pointer(a), C.pointer(b)
        a.parent = b.parent = C.pointer(p)
        inner += 1
        active.append(p)
    return nodes, active[0]


def run_rtree(lib, root, trav, n_nodes):
    CB = C.CFUNCTYPE(C.c_int, C.POINTER(RNode))
    cb = CB(lambda p: 1)
    buf = (C.POINTER(RNode) * n_nodes)()
    size = C.c_uint()
    lib.pll_rtree_traverse.argtypes = [C.POINTER(RNode), C.c_int, CB, C.POINTER(C.POINTER(RNode)),
                                       C.POINTER(C.c_uint)]
    rc = lib.pll_rtree_traverse(C.pointer(root), trav, cb, buf, C.byref(size))
    order = [buf[i].contents.clv_index for i in range(size.value)]
    branches = (C.c_double * n_nodes)()
    pmi = (C.c_uint * n_nodes)()
    ops = (Operation * n_nodes)()
    mc, oc = C.c_uint(), C.c_uint()
    lib.pll_rtree_create_operations.restype = None
    lib.pll_rtree_create_operations.argtypes = [C.POINTER(C.POINTER(RNode)), C.c_uint,
                                                C.POINTER(C.c_double), C.POINTER(C.c_uint),
                                                C.POINTER(Operation), C.POINTER(C.c_uint),
                                                C.POINTER(C.c_uint)]
    lib.pll_rtree_create_operations(buf, size, branches, pmi, ops, C.byref(mc), C.byref(oc))
    ops_t = [tuple(getattr(ops[i], f) for f, _ in Operation._fields_) for i in range(oc.value)]
    return rc, order, list(branches[:mc.value]), list(pmi[:mc.value]), ops_t


@pytest.mark.parametrize("tips", [2, 3, 9, 120])
def test_rtree_traverse_and_operations(amd, ref_tree, tips):
    ref = ref_tree
    rng = np.random.default_rng(100 + tips)
    nodes, root = random_rtree(tips, rng)
    for trav in (1, 2):
        assert run_rtree(amd.lib, root, trav, len(nodes)) == run_rtree(ref.lib, root, trav, len(nodes))


def test_deep_caterpillar_does_not_overflow_the_stack(amd):
    """100 000-tip ladder: the explicit-stack walk handles what recursion could not."""
    tips = 100_000
    nodes = (RNode * (2 * tips - 1))()
    for i in range(tips):
        nodes[i].clv_index = i
    prev = 0
    for k in range(tips - 1):
        p = tips + k
        nodes[p].clv_index = p
        nodes[p].left = C.pointer(nodes[prev])
        nodes[p].right = C.pointer(nodes[k + 1])
        prev = p
    CB = C.CFUNCTYPE(C.c_int, C.POINTER(RNode))
    buf = (C.POINTER(RNode) * (2 * tips - 1))()
    size = C.c_uint()
    amd.lib.pll_rtree_traverse.argtypes = [C.POINTER(RNode), C.c_int, CB,
                                           C.POINTER(C.POINTER(RNode)), C.POINTER(C.c_uint)]
    keep = C.cast(amd.lib.pll_amd_accept_all_rnodes, CB) if hasattr(amd.lib, "pll_amd_accept_all_rnodes") \
        else CB(lambda p: 1)
    assert amd.lib.pll_rtree_traverse(C.pointer(nodes[prev]), 1, keep, buf, C.byref(size)) == 1
    assert size.value == 2 * tips - 1 and buf[size.value - 1].contents.clv_index == prev


def compress(lib, seqs, cmap):
    n = len(seqs[0])
    bufs = [C.create_string_buffer(s, n + 1) for s in seqs]
    arr = (C.c_char_p * len(seqs))(*[C.cast(b, C.c_char_p) for b in bufs])
    length = C.c_int(n)
    cmap = np.ascontiguousarray(cmap, dtype=np.uint32)
    lib.pll_compress_site_patterns.restype = C.POINTER(C.c_uint)
    lib.pll_compress_site_patterns.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.c_int,
                                               C.POINTER(C.c_int)]
    w = lib.pll_compress_site_patterns(arr, cmap.ctypes.data_as(C.POINTER(C.c_uint)), len(seqs),
                                       C.byref(length))
    if not w:
        return None
    weights = [w[i] for i in range(length.value)]
    return length.value, weights, [b.raw[:length.value + 1] for b in bufs]


@pytest.mark.parametrize("states,taxa,sites", [(4, 5, 40), (4, 12, 3000), (20, 7, 500), (4, 1, 10),
                                               (4, 30, 1)])
def test_compress_site_patterns(amd, ref, states, taxa, sites):
    rng = np.random.default_rng(sites + taxa)
    alphabet = np.frombuffer(b"ACGTacgtRYN-?" if states == 4 else b"ARNDCQEGHILKMFPSTWYVBZX-arnd",
                             dtype=np.uint8)
    # few distinct columns -> many repeats
    pool = alphabet[rng.integers(0, len(alphabet), size=(max(2, sites // 6), taxa))]
    cols = pool[rng.integers(0, len(pool), size=sites)]
    seqs = [cols[:, t].tobytes() for t in range(taxa)]
    cmap = amd.map("nt" if states == 4 else "aa")
    a = compress(amd.lib, seqs, cmap)
    r = compress(ref.lib, seqs, cmap)
    assert a == r
    assert sum(a[1]) == sites and a[0] <= sites


def test_compress_rejects_bad_arguments(amd, ref):
    cmap = amd.map("nt").copy()
    cmap[0] = 1           # a state for NUL is not allowed (compress.c:153)
    assert compress(amd.lib, [b"ACGT"], cmap) is None and compress(ref.lib, [b"ACGT"], cmap) is None


def test_compress_remaps_wide_state_codes(amd, ref):
    """Maps whose values exceed a byte (20-bit amino-acid masks) are renumbered."""
    rng = np.random.default_rng(5)
    chars = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV-", dtype=np.uint8)
    seqs = [chars[rng.integers(0, len(chars), 300)].tobytes() for _ in range(6)]
    assert compress(amd.lib, seqs, amd.map("aa")) == compress(ref.lib, seqs, ref.map("aa"))


def test_identify_repeats(amd):
    """Class identification of the site-repeats extension (host/repeats.c): distinct
    pairs numbered by first appearance, both table strategies, and the give-up limit."""
    import ctypes as C
    f = amd.lib.pll_amd_identify_repeats
    f.restype = C.c_uint
    rng = np.random.default_rng(3)
    for na, nb, sites in ((16, 16, 5000), (300, 7, 4000), (5000, 3000, 20000), (1, 1, 10)):
        ida = rng.integers(0, na, sites).astype(np.uint32)
        idb = rng.integers(0, nb, sites).astype(np.uint32)
        sid = np.zeros(sites, dtype=np.uint32)
        lrow = np.zeros(sites, dtype=np.uint32)
        rrow = np.zeros(sites, dtype=np.uint32)
        def ptr(a):
            return a.ctypes.data_as(C.POINTER(C.c_uint))
        n = f(ptr(ida), na, ptr(idb), nb, sites, sites, ptr(sid), ptr(lrow), ptr(rrow))
        keys = ida.astype(np.uint64) * nb + idb
        uniq, first = np.unique(keys, return_index=True)
        assert n == len(uniq)
        order = np.argsort(first)                       # classes in order of first appearance
        rank = np.empty(len(uniq), dtype=np.uint32)
        rank[order] = np.arange(len(uniq), dtype=np.uint32)
        assert (sid == rank[np.searchsorted(uniq, keys)]).all()
        assert (lrow[:n] == ida[np.sort(first)]).all() and (rrow[:n] == idb[np.sort(first)]).all()
        assert (sid <= np.arange(sites)).all()           # what the in-place expansion relies on
        if n > 1:
            assert f(ptr(ida), na, ptr(idb), nb, sites, n - 1, ptr(sid), ptr(lrow), ptr(rrow)) == 0
