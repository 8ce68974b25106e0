#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the small summaries kept under profiles/.

  kernel trace  (rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py ...)
      summarize_rocprof.py stats DIR OUT.csv
      copies *_kernel_stats.csv (one line per kernel: calls, total, average, min, max)

  PMC passes    (rocprofv3 --pmc FETCH_SIZE -d DIR_F ... ; rocprofv3 --pmc WRITE_SIZE -d DIR_W ...)
      summarize_rocprof.py hbm DIR_F DIR_W OUT.csv ["command line"]
      per kernel: mean FETCH_SIZE / WRITE_SIZE (KB) and HBM MB per launch
          = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 / 1e6
      (gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes, WRITE_SIZE is exact:
      /opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section)

  generic PMC   summarize_rocprof.py pmc DIR OUT.csv ["command line"]
      per kernel and counter: dispatches, mean, sum

  traffic index summarize_rocprof.py index SPEC.json OUT.json
      SPEC: list of {"csv": hbm summary, "kernel_match": substring of the kernel name,
      "kernel_class": "whole-list" | "inner-inner", "workload": {states, rate_cats, sites, taxa,
      tree, tip_clv, rate_scalers}, "bench_json": optional bench line of the same command (its
      roofline.ops_per_launch is recorded)} -> the list bench.py looks `roofline.traffic` up in
"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        sys.exit("no *%s under %s" % (suffix, d))
    return hits


def counters(d):
    """{kernel: {counter: [values per dispatch]}} in first-seen order."""
    out = OrderedDict()
    for path in find(d, "_counter_collection.csv"):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = out.setdefault(row["Kernel_Name"], defaultdict(list))
                k[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def mean(v):
    return sum(v) / len(v) if v else 0.0


def cmd_stats(d, out):
    lines, seen_header = [], False
    for path in find(d, "_kernel_stats.csv"):
        for line in open(path):
            if line.startswith('"Name"'):
                if seen_header:
                    continue
                seen_header = True
            lines.append(line)
    with open(out, "w") as f:
        f.writelines(lines)


def cmd_hbm(df, dw, out, command=""):
    fetch, write = counters(df), counters(dw)
    with open(out, "w") as f:
        if command:
            f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: %s\n" % command)
        f.write("# hbm_MB_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 / 1e6  "
                "(gfx950: FETCH_SIZE counts 128-B requests at 64 B, MI355X_MICROARCH.md)\n")
        f.write("# batched launches (blockIdx.y = op) carry several ops per launch\n")
        f.write("kernel,dispatches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_MB_per_launch\n")
        for k in sorted(set(fetch) | set(write)):
            fv = fetch.get(k, {}).get("FETCH_SIZE", [])
            wv = write.get(k, {}).get("WRITE_SIZE", [])
            fm, wm = mean(fv), mean(wv)
            f.write('"%s",%d,%.1f,%.1f,%.2f\n' % (k, max(len(fv), len(wv)), fm, wm,
                                                 (2 * fm + wm) * 1024 / 1e6))


def cmd_pmc(d, out, command=""):
    c = counters(d)
    with open(out, "w") as f:
        if command:
            f.write("# rocprofv3 --pmc ... of: %s\n" % command)
        f.write("kernel,counter,dispatches,mean,sum\n")
        for k in sorted(c):
            for name in sorted(c[k]):
                v = c[k][name]
                f.write('"%s",%s,%d,%.3f,%.3f\n' % (k, name, len(v), mean(v), sum(v)))


def cmd_index(spec_path, out):
    import json
    spec = json.load(open(spec_path))
    base = os.path.dirname(os.path.abspath(out))
    index = []
    for e in spec:
        path = e["csv"] if os.path.isabs(e["csv"]) else os.path.join(base, e["csv"])
        if not os.path.exists(path):
            continue
        best = None
        with open(path, newline="") as f:
            rows = [r for r in csv.reader(l for l in f if not l.startswith("#"))]
        for r in rows[1:]:
            if e["kernel_match"] in r[0] and (best is None or float(r[4]) > float(best[4])):
                best = r
        if best is None:
            continue
        mb = float(best[4])
        name = best[0]
        if e.get("sum_call"):
            # a call that is several launches (20 states: tip-tip ops and tables ahead of the list kernel):
            # everything the call launches, per launch of the kernel that runs once per call
            calls = int(best[1])
            skip = e.get("exclude", [])
            mb = sum(float(r[4]) * int(r[1]) for r in rows[1:] if not any(x in r[0] for x in skip)) / calls
            name = "every launch of the call (per launch of %s)" % best[0][:60]
        entry = {"workload": dict(e["workload"], kernel_class=e["kernel_class"]), "kernel": name,
                 "dispatches": int(best[1]), "hbm_MB_per_launch": mb,
                 "source": "profiles/" + os.path.basename(path)}
        mc = e.get("mfma_csv")
        if mc:
            # SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) against the launch's duration in shader-engine cycles:
            # GRBM_GUI_ACTIVE is counted per XCD (8), a SIMD's share of the chip is 1 / 1024
            mc = mc if os.path.isabs(mc) else os.path.join(base, mc)
            try:
                with open(mc, newline="") as f:
                    mrows = [r for r in csv.reader(l for l in f if not l.startswith("#"))]
                val = {r[1]: float(r[3]) for r in mrows[1:] if e["kernel_match"] in r[0]}
                entry["mfma_busy_frac"] = round(val["SQ_VALU_MFMA_BUSY_CYCLES"] / (val["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
                entry["mfma_source"] = "profiles/" + os.path.basename(mc)
            except (OSError, KeyError, ValueError, ZeroDivisionError):
                pass
        bj = e.get("bench_json")
        if bj:
            bj = bj if os.path.isabs(bj) else os.path.join(base, bj)
            try:
                line = [l for l in open(bj).read().splitlines() if l.startswith("{")][-1]
                entry["ops_per_launch"] = json.loads(line)["roofline"].get("ops_per_launch")
            except (OSError, ValueError, IndexError, KeyError):
                pass
        index.append(entry)
    with open(out, "w") as f:
        json.dump(index, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    a = sys.argv[1:]
    if len(a) >= 3 and a[0] == "stats":
        cmd_stats(a[1], a[2])
    elif len(a) >= 4 and a[0] == "hbm":
        cmd_hbm(a[1], a[2], a[3], a[4] if len(a) > 4 else "")
    elif len(a) >= 3 and a[0] == "index":
        cmd_index(a[1], a[2])
    elif len(a) >= 3 and a[0] == "pmc":
        cmd_pmc(a[1], a[2], a[3] if len(a) > 3 else "")
    else:
        sys.exit(__doc__)
