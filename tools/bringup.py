#!/usr/bin/env python3
"""Bring-up check: run the same call sequence on the product (HIP) and on the
reference CPU library and print the differences.  Developer tool; the real
parity suite is tests/."""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import *

amd = libpll_amd.load()
ref = PllLibrary(os.path.join(root, "oracle", "_ref", "libpll_ref.so"))
print("devices:", amd.device_count())

def run(lib, plan, seqs, states, cats, attrs, pw=None, pinv=0.0):
    p = W.setup_partition(lib, plan, seqs, states, cats, attrs, pattern_weights=pw, pinv=pinv)
    p.update_partials(plan.ops)
    lnl, ps = p.compute_edge_loglikelihood(*plan.root_edge, [0]*cats, persite=True)
    return p, lnl, ps

def cmp(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.dtype.kind == 'f':
        ne = int((a.view(np.uint64) != b.view(np.uint64)).sum()) if a.shape==b.shape else -1
        with np.errstate(all='ignore'):
            rel = np.nanmax(np.abs(a-b)/np.maximum(np.abs(b),1e-300)) if a.size else 0
        return "%s: %d/%d bits differ, maxrel %.3g" % (name, ne, a.size, rel)
    return "%s: %d/%d differ" % (name, int((a!=b).sum()), a.size)

for states in (4, 20):
  for shape, T, sites in (("balanced", 16, 1000), ("caterpillar", 300, 24), ("random", 23, 333)):
    for pt in (0, ATTRIB_PATTERN_TIP):
      for rs in (0, ATTRIB_RATE_SCALERS):
        plan = getattr(W, shape + "_tree")(T, seed=7)
        seqs = W.random_alignment(T, sites, states, seed=3)
        pw = np.random.default_rng(5).integers(1, 4, size=sites).astype(np.uint32)
        t0 = time.time()
        pa, la, psa = run(amd, plan, seqs, states, 4, pt | rs, pw)
        pr, lr, psr = run(ref, plan, seqs, states, 4, pt | rs | ATTRIB_ARCH_AVX2, pw)
        msgs = []
        worst_clv = 0; nclvbits = 0; nsc = 0
        for m in plan.matrix_indices[:3]:
            msgs.append(cmp("P%d" % m, pa.get_pmatrix(int(m)), pr.get_pmatrix(int(m))))
        for op in plan.ops:
            ca = pa.get_clv(int(op["parent_clv_index"])); cr = pr.get_clv(int(op["parent_clv_index"]))
            nclvbits += int((ca.view(np.uint64) != cr.view(np.uint64)).sum())
            with np.errstate(all='ignore'):
                worst_clv = max(worst_clv, float(np.nanmax(np.abs(ca-cr)/np.maximum(np.abs(cr),1e-300))))
            si = int(op["parent_scaler_index"])
            nsc += int((pa.get_scaler(si) != pr.get_scaler(si)).sum())
        last = int(plan.ops[-1]["parent_scaler_index"])
        print("S=%d %s T=%d pt=%d rs=%d | lnL hip %.12f ref %.12f rel %.2e | clv bits %d maxrel %.2e | scaler diffs %d (max scaler %d) | %s | %s | %.1fs"
              % (states, shape, T, bool(pt), bool(rs), la, lr, abs(la-lr)/abs(lr), nclvbits, worst_clv, nsc,
                 int(pr.get_scaler(last).max()), cmp("persite", psa, psr), "; ".join(msgs), time.time()-t0))
        # derivatives at the root edge
        e = plan.root_edge
        sa = pa.alloc_sumtable(); sr = pr.alloc_sumtable()
        pa.update_sumtable(e[0], e[2], e[1], e[3], [0]*4, sa); pr.update_sumtable(e[0], e[2], e[1], e[3], [0]*4, sr)
        da = pa.compute_likelihood_derivatives(e[1], e[3], 0.13, [0]*4, sa)
        dr = pr.compute_likelihood_derivatives(e[1], e[3], 0.13, [0]*4, sr)
        print("    ", cmp("sumtable", pa.get_sumtable(sa), pr.get_sumtable(sr)), "| d %.12g/%.12g dd %.12g/%.12g" % (da[0], dr[0], da[1], dr[1]))
        pa.destroy(); pr.destroy()
