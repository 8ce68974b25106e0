#!/usr/bin/env python3
"""developer tool: what a NEW 20-state op list costs on the device, kernel by kernel.
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/new_list_trace.py run [states] [sites]
  python3 tools/new_list_trace.py parse DIR
`run` hands pll_update_partials, for each of five traversal roots, another list, then the root's list (new), then the
same list three times (kept plan), four trials; `parse` lines the dispatches up in time order and prints per root the
median duration of k_af_prepare / the list kernel and the idle time in front of them for the new call and for the replays."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(states, sites):
    import time
    import libpll_amd
    from libpll_amd import workload as W
    from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
    lib = libpll_amd.load()
    plan = W.balanced_tree(64, seed=42)
    R = 4
    cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
    rates, freqs = (W.GTR_RATES, W.GTR_FREQS) if states == 4 else lib.aa_model("lg")
    seqs = W.simulated_alignment(plan, sites, rates, freqs, cats, seed=42)
    p = W.setup_partition(lib, plan, seqs, states, R, ATTRIB_PATTERN_TIP)
    view = W.UnrootedView(plan)
    rng = W.SplitMix64(777)
    inner = [e for e in view.edges() if e[0] >= 64 and e[1] >= 64]
    roots = [view.root] + [inner[rng.below(len(inner))] for _ in range(4)]
    for _ in range(20):
        p.update_partials(plan.ops)
    p.wait()
    other = view.traversal(roots[1])[0]
    for r in roots:
        ops, edge = view.traversal(r)
        for trial in range(4):
            p.update_partials(plan.ops if r != view.root else other)
            p.wait()
            time.sleep(0.002)
            for k in range(4):
                t0 = time.perf_counter()
                p.update_partials(ops)
                t1 = time.perf_counter()
                p.wait()
                t2 = time.perf_counter()
                print("root %s call %d: returns after %.1f us, done after %.1f us" % (r, k, (t1 - t0) * 1e6, (t2 - t0) * 1e6))
                time.sleep(0.002)
    p.destroy()


def parse(d):
    import csv
    import glob
    import re
    import numpy as np
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            m = re.search(r"k_\w+", row["Kernel_Name"])
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), m.group(0) if m else row["Kernel_Name"][:40]))
    for path in glob.glob(os.path.join(d, "**", "*_memory_copy_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), "copy " + row.get("Direction", "")))
    rows.sort()
    # a call = everything between two list kernels
    calls, cur = [], []
    for s, e, name in rows:
        cur.append((s, e, name))
        if "fused" in name:
            calls.append(cur)
            cur = []
    calls = calls[20:]  # the warm-up evaluations
    # per root: 4 trials x (other, new, replay x 3)
    per = 4 * 5
    for ri in range(len(calls) // per):
        new, rep = [], []
        for t in range(4):
            grp = calls[ri * per + t * 5: ri * per + t * 5 + 5]
            for k, c in enumerate(grp[1:]):
                d_list = (c[-1][1] - c[-1][0]) / 1e3
                prep = [x for x in c if "prepare" in x[2]]
                d_prep = sum(x[1] - x[0] for x in prep) / 1e3
                gap = (c[-1][0] - prep[-1][1]) / 1e3 if prep else float("nan")
                span = (c[-1][1] - c[0][0]) / 1e3
                others = ", ".join("%s %.1f" % (x[2], (x[1] - x[0]) / 1e3) for x in c[:-1] if "prepare" not in x[2])
                (new if k == 0 else rep).append((d_prep, gap, d_list, span, others))
        fmt = lambda v: "prepare %6.1f us, gap %5.1f, list kernel %8.1f, first dispatch to end %8.1f" % tuple(np.median(np.array([x[:4] for x in v]), axis=0))
        print("root %d  new:    %s   [%s]" % (ri, fmt(new), new[0][4]))
        print("        replay: %s   [%s]" % (fmt(rep), rep[0][4]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 200000)
    else:
        parse(sys.argv[2])
