#!/usr/bin/env python3
"""developer tool (round 6): hardware counters of the 20-state list kernel by what ran before it.
  rocprofv3 --pmc <counters> --output-format csv -d DIR -- python3 tools/replay_pmc.py run
  python3 tools/replay_pmc.py parse DIR [DIR ...]
`run`: for traversal roots 0 and 1 of BASELINE config 3's partition, three times:
  another root's list / this root's list (new) / again (replay) / 1 GiB of unrelated memory overwritten / again
  (replay+flush) / again (replay)
`parse`: the k_aa_fused dispatches in order, labelled, counters averaged per label and root."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LABELS = ["other", "new", "replay", "replay+flush", "replay again"]
ROOTS = tuple(int(x) for x in os.environ.get("REPLAY_ROOTS", "0,1").split(","))
TRIALS = 3


def run():
    os.environ.setdefault("PLLHIP_DEVELOPER", "1")
    import torch
    import libpll_amd
    from libpll_amd import workload as W
    from libpll_amd.pllapi import ATTRIB_PATTERN_TIP
    lib = libpll_amd.load()
    taxa, sites, R = 64, 200_000, 4
    plan = W.balanced_tree(taxa, seed=42)
    cats = lib.compute_gamma_cats(W.GAMMA_ALPHA, R)
    rates, freqs = lib.aa_model("lg")
    seqs = W.simulated_alignment(plan, sites, rates, freqs, cats, seed=42)
    p = W.setup_partition(lib, plan, seqs, 20, R, ATTRIB_PATTERN_TIP)
    view = W.UnrootedView(plan)
    rng = W.SplitMix64(777)
    inner = [e for e in view.edges() if e[0] >= taxa and e[1] >= taxa]
    roots = [view.root] + [inner[rng.below(len(inner))] for _ in range(4)]
    lists = [view.traversal(r)[0] for r in roots]
    scratch = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    # (no warm-up launches of the list kernel: every k_aa_fused dispatch of the process is one of the labelled ones)
    for i in ROOTS:
        for _ in range(TRIALS):
            p.update_partials(lists[(i + 1) % len(lists)]); p.wait()
            p.update_partials(lists[i]); p.wait()
            p.update_partials(lists[i]); p.wait()
            scratch.fill_(1.0); torch.cuda.synchronize()
            p.update_partials(lists[i]); p.wait()
            p.update_partials(lists[i]); p.wait()
    p.destroy()


def parse(dirs):
    import csv, glob
    from collections import defaultdict, OrderedDict
    for d in dirs:
        rows = []
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if "k_aa_fused" in row["Kernel_Name"]:
                    rows.append((int(row["Dispatch_Id"]), row["Counter_Name"], float(row["Counter_Value"])))
        ids = sorted(set(r[0] for r in rows))
        label_of = {}
        for n, did in enumerate(ids):
            per_root = len(LABELS) * TRIALS
            label_of[did] = (ROOTS[min(n // per_root, len(ROOTS) - 1)], LABELS[n % len(LABELS)])
        acc = OrderedDict()
        for did, name, val in rows:
            acc.setdefault(label_of[did], defaultdict(list))[name].append(val)
        names = sorted(set(r[1] for r in rows))
        print("%s: %d k_aa_fused dispatches" % (d, len(ids)))
        print("%-24s" % "root, launch" + "".join("%26s" % n[:25] for n in names))
        for key, vals in acc.items():
            print("%-24s" % ("%d %s" % key) + "".join("%26.4g" % (sum(vals[n]) / len(vals[n])) for n in names))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) >= 3 and sys.argv[1] == "parse":
        parse(sys.argv[2:])
    else:
        sys.exit(__doc__)
