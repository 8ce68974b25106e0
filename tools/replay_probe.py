#!/usr/bin/env python3
"""developer tool (round 6): why does the 20-state list kernel run ~7 % slower when an op list
follows ITSELF than when it follows another list (two of five traversal roots, profiles/r5_new_list_kernel_trace_c3.txt)?
HIP-event time of pll_update_partials(list) on the partition's stream, by what ran before it:
  new            another root's list, then this one (planned anew)
  replay         this list again (kept plan)
  replay+flush   this list again after 1 GiB of unrelated memory was overwritten (whatever the last launch left in
                 L2 / the memory-side cache is gone)
  new+flush      another root's list, the same flush, then this one
  replay x8      eight replays back to back, per-call average
  python3 tools/replay_probe.py [states] [sites] [taxa]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("PLLHIP_DEVELOPER", "1")
import numpy as np
import torch
import libpll_amd
from libpll_amd import workload as W
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP

states = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
taxa = int(sys.argv[3]) if len(sys.argv) > 3 else 64
lib = libpll_amd.load()
plan = W.balanced_tree(taxa, seed=42)
R = 4
cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
rates, freqs = (W.GTR_RATES, W.GTR_FREQS) if states == 4 else lib.aa_model("lg")
seqs = W.simulated_alignment(plan, sites, rates, freqs, cats, seed=42)
p = W.setup_partition(lib, plan, seqs, states, R, ATTRIB_PATTERN_TIP)
view = W.UnrootedView(plan)
rng = W.SplitMix64(777)
inner = [e for e in view.edges() if e[0] >= taxa and e[1] >= taxa]
roots = [view.root] + [inner[rng.below(len(inner))] for _ in range(4)]
scratch = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB


def flush():
    scratch.fill_(1.0)
    torch.cuda.synchronize()


def timed(ops, n=1):
    p.wait()
    p.timer_start()
    for _ in range(n):
        p.update_partials(ops)
    return p.timer_stop_ms() * 1e3 / n


for _ in range(20):
    p.update_partials(plan.ops)
p.wait()
lists = [view.traversal(r)[0] for r in roots]
print("%d states, %d sites, %d taxa: us per pll_update_partials call (HIP events), median of 7" % (states, sites, taxa))
for i, ops in enumerate(lists):
    other = lists[(i + 1) % len(lists)]
    res = {k: [] for k in ("new", "replay", "replay+flush", "new+flush", "replay x8", "replay after 2 others",
                           "replay, other branch lengths", "replay again", "replay after the bare stores")}
    for trial in range(7):
        p.update_partials(other); p.wait()
        res["new"].append(timed(ops))
        res["replay"].append(timed(ops))
        flush()
        res["replay+flush"].append(timed(ops))
        p.update_partials(other); p.wait(); flush()
        res["new+flush"].append(timed(ops))
        res["replay x8"].append(timed(ops, 8))
        p.update_partials(other); p.update_partials(lists[(i + 2) % len(lists)]); p.wait()
        res["replay after 2 others"].append(timed(ops))
        # the same list, the same addresses -- other VALUES: every branch 0.1 % longer
        p.update_partials(ops); p.wait()
        p.update_prob_matrices([0] * R, plan.matrix_indices, plan.branch_lengths * 1.001)
        res["replay, other branch lengths"].append(timed(ops))
        res["replay again"].append(timed(ops))
        p.update_prob_matrices([0] * R, plan.matrix_indices, plan.branch_lengths)
        p.update_partials(ops); p.wait()
        # the same addresses written by another kernel (ones everywhere), then the list
        p.write_ceiling(ops, 1)
        res["replay after the bare stores"].append(timed(ops))
    print("root %d (%s):" % (i, roots[i]), "  ".join("%s %.0f" % (k, float(np.median(v))) for k, v in res.items()))
p.destroy()
