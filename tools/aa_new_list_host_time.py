import os, sys, time
sys.path.insert(0, "."); 
import numpy as np
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
amd = libpll_amd.load()
T, sites, R = 64, 200_000, 4
plan = W.balanced_tree(T, seed=42)
rates, freqs = amd.aa_model("lg")
seqs = W.global_alignment(plan, 0, sites, rates, freqs, amd.compute_gamma_cats(W.GAMMA_ALPHA, R), seed=42)
p = W.setup_partition(amd, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
view = W.UnrootedView(plan)
edges = [e for e in view.edges() if e[0] >= T and e[1] >= T]
lists = [view.traversal(e) for e in edges[:6]]
for ops, edge in lists:            # warm: buffers grow
    p.update_partials(ops); p.compute_edge_loglikelihood(*edge, [0]*R)
p.wait()
os.environ["PLLHIP_FUSED_DEBUG"] = "3"
for k, (ops, edge) in enumerate(lists[:3]):
    t = time.perf_counter(); p.update_partials(ops); t1 = time.perf_counter(); p.wait(); t2 = time.perf_counter()
    print("new list %d: call returns after %.1f us, done after %.1f us" % (k, (t1-t)*1e6, (t2-t)*1e6), file=sys.stderr)
del os.environ["PLLHIP_FUSED_DEBUG"]
ops, edge = lists[0]
p.update_partials(ops); p.wait()
for k in range(3):
    t = time.perf_counter(); p.update_partials(ops); t1 = time.perf_counter(); p.wait(); t2 = time.perf_counter()
    print("same list again: call returns after %.1f us, done after %.1f us" % ((t1-t)*1e6, (t2-t)*1e6), file=sys.stderr)
