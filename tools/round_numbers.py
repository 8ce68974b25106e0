#!/usr/bin/env python3
"""developer tool: the numbers DESIGN.md section 4 quotes, read back from a round's committed profiles
(profiles/<tag>_bench_*.json, *_kernel_stats.csv).   python3 tools/round_numbers.py [tag] [dir]"""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
d = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for f in sorted(glob.glob(os.path.join(d, tag + "_bench_*.json"))):
    name = os.path.basename(f)[len(tag) + 7:-5]
    if name.endswith("_under_rocprof"):
        continue
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "unreadable:", e)
        continue
    r = b.get("roofline") or {}
    bc = r.get("box_ceiling") or {}
    mc = r.get("matrix_cores") or {}
    cb = b.get("cpu_baseline") or {}
    line = "%-28s value %9.1f  ms/step %7.3f  frac %.3f  launch %8.1f us  traffic %s MB" % (
        name, b["value"], b["ms_per_step"], r.get("frac", 0), r.get("avg_launch_us", 0),
        ("%.0f" % (r["traffic"] / 1e6)) if r.get("traffic") else "-")
    if bc:
        line += "  ceiling %.1f GB/s (%.3f of peak), of ceiling %.3f" % (bc["GBs"], bc["frac_of_peak"], r.get("frac_of_box_ceiling", 0))
    if mc:
        line += "  mfma_busy %s, %.2f TF on matrix cores" % (mc.get("mfma_busy"), mc.get("tflops_on_matrix_cores", 0))
    if cb:
        line += "  cpu %.1f / %.1f M/s (%s cores)" % (cb.get("one_core_value", 0), cb.get("value", 0), cb.get("cores"))
    if b.get("lnl_rel_err_vs_reference") is not None:
        line += "  lnL err %.1e" % b["lnl_rel_err_vs_reference"]
    c4 = b.get("c4_strong")
    if c4:
        line += "  c4_strong %s" % json.dumps({k: c4[k] for k in c4 if k in ("value", "ms_per_step", "lnl_rel_err_vs_reference", "sites_checked")})
    vl = b.get("varying_lists")
    if vl:
        line += "  varying %s" % json.dumps(vl)[:300]
    print(line)
    k = os.path.join(d, "%s_bench_%s_kernel_stats.csv" % (tag, name))
    if os.path.exists(k):
        for row in csv.DictReader(open(k)):
            n = row["Name"]
            short = next((s for s in ("k_dna_fused", "k_aa_fused", "k_af_prepare", "k_dna_pair_tables", "k_write_ceiling", "k_lnl_dna", "k_lnl_aa_mfma",
                                      "k_dna_partials", "k_aa_ii_mfma", "k_aa_tt_rounds", "k_aa_cherry_rounds", "k_derivatives", "k_sumtable") if s in n), None)
            if short:
                print("      %-20s calls %6s  average %9.1f us" % (short, row["Calls"], float(row["AverageNs"]) / 1e3))
for f in ("size_sweep.txt", "result_calls_from_c.txt"):
    p = os.path.join(d, tag + "_" + f)
    if os.path.exists(p):
        print("==", f)
        print(open(p).read().rstrip()[:3000])
