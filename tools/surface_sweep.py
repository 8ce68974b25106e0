#!/usr/bin/env python3
"""Effective HBM rate of every call of the hot path over the shapes the API admits -- a search for kernels that are far
from the roofline on shapes no BASELINE config names (round 4: 20 states with 3, 6, 16 rate categories at a third of
the 4-category rate on every call; profiles/r4_surface_sweep*.txt).  Run on the GPU box:  python3 tools/surface_sweep.py [--quick]

Per shape: a 32-taxon random tree, sites chosen so that a CLV is ~64 MB; three evaluations (update_partials + edge
lnL), one root lnL, one sumtable, five derivative calls, timed per kernel class with the library's own HIP-event
profile (pll_amd_profile_*).  Rates are ALGORITHMIC bytes / time: 8 * states * rate_cats bytes per CLV row read or
written (children that are tip characters: 1 byte), sumtable = a CLV row.
"""
import argparse
import os
os.environ.setdefault("PLL_AMD_AUTO_MIRROR_MB", "0")   # (the device path is what is measured: no host mirrors kept)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libpll_amd  # noqa: E402
from helpers import make_case, odd_state_case, many_state_case, build_partition  # noqa: E402
from libpll_amd.pllapi import ATTRIB_PATTERN_TIP, ATTRIB_RATE_SCALERS, ATTRIB_SITE_REPEATS  # noqa: E402



def run(lib, states, rate_cats, attrs, tip_clv, taxa=32, target_mb=64, pinv=0.0):
    row = 8 * states * rate_cats
    sites = max(2000, int(target_mb * 1e6 / row))
    kw = dict(tips=taxa, sites=sites, seed=7, rate_cats=rate_cats)
    if states in (4, 20):
        case = make_case(states, "random", **kw)
        if states == 20:
            case["rates"], case["freqs"] = lib.aa_model("lg")
    elif states > 32:
        case = many_state_case(states, **kw)
    else:
        case = odd_state_case(states, **kw)
    if tip_clv or states > 32:
        attrs &= ~ATTRIB_PATTERN_TIP
    p = build_partition(lib, case, attrs, pinv=pinv)
    plan = case["plan"]
    R = rate_cats
    e = plan.root_edge
    # warm
    p.update_partials(plan.ops)
    p.compute_edge_loglikelihood(*e, [0] * R)
    st = p.alloc_sumtable()
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)   # (a kernel's first launch in a process loads its code)
    p.compute_likelihood_derivatives(e[1], e[3], 0.2, [0] * R, st)
    p.profile_enable(True)
    p.profile_read()
    for _ in range(3):
        p.update_partials(plan.ops)
        p.compute_edge_loglikelihood(*e, [0] * R)
    p.compute_root_loglikelihood(e[0], e[1], [0] * R)
    p.update_sumtable(e[0], e[2], e[1], e[3], [0] * R, st)
    for t in (0.01, 0.1, 0.3, 0.5, 1.0):
        p.compute_likelihood_derivatives(e[1], e[3], t, [0] * R, st)
    prof = p.profile_read()
    p.destroy()
    tt, ti, ii = plan.op_kinds() if (attrs & ATTRIB_PATTERN_TIP) else (0, 0, len(plan.ops))
    part_bytes = 3 * sites * (ii * 3 * row + ti * (2 * row + 1) + tt * (row + 2))
    part_ms = sum(prof[k][1] for k in ("partials_ii", "partials_ti", "partials_tt"))
    out = {"partials": part_bytes / part_ms / 1e6 if part_ms else float("nan")}
    n, ms = prof["lnl"]
    out["lnl"] = n * sites * 2 * row / ms / 1e6 if ms else float("nan")
    out["lnl_us"] = ms / max(n, 1) * 1e3
    n, ms = prof["sumtable"]
    out["sumtable"] = n * sites * 3 * row / ms / 1e6 if ms else float("nan")
    n, ms = prof["derivatives"]
    out["derivatives"] = n * sites * row / ms / 1e6 if ms else float("nan")
    out["deriv_us"] = ms / max(n, 1) * 1e3
    out["sites"] = sites
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--floor", type=float, default=1500.0, help="flag rates below this many GB/s")
    a = ap.parse_args()
    lib = libpll_amd.load()
    shapes = []
    for S in (4, 20):
        for R in ((4, 8) if a.quick else (1, 2, 3, 4, 6, 8, 16)):
            for tip_clv in (False, True):
                for rs in (0, ATTRIB_RATE_SCALERS):
                    shapes.append((S, R, ATTRIB_PATTERN_TIP | rs, tip_clv, 0.0))
        shapes.append((S, 4, ATTRIB_PATTERN_TIP, False, 0.2))
        shapes.append((S, 4, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS, False, 0.0))
        shapes.append((S, 8, ATTRIB_PATTERN_TIP | ATTRIB_SITE_REPEATS, False, 0.0))
    if not a.quick:
        for S in (2, 3, 5, 7, 13, 16, 24, 32, 61):
            for R in (1, 4):
                shapes.append((S, R, ATTRIB_PATTERN_TIP, False, 0.0))
    print("%-7s %-4s %-8s %-6s %-5s %-4s | %9s %9s %9s %9s | %8s %8s" % (
        "states", "R", "tips", "scaler", "pinv", "rep", "partials", "lnl", "sumtable", "deriv", "lnl_us", "deriv_us"))
    for S, R, attrs, tip_clv, pinv in shapes:
        try:
            o = run(lib, S, R, attrs, tip_clv, pinv=pinv)
        except Exception as ex:  # noqa: BLE001
            print("%-7d %-4d FAILED: %s" % (S, R, ex))
            continue
        flags = [k for k in ("partials", "lnl", "sumtable", "derivatives") if o[k] < a.floor]
        print("%-7d %-4d %-8s %-6s %-5.2f %-4s | %9.0f %9.0f %9.0f %9.0f | %8.1f %8.1f   %s" % (
            S, R, "clv" if tip_clv else "chars", "rate" if attrs & ATTRIB_RATE_SCALERS else "site", pinv,
            "yes" if attrs & ATTRIB_SITE_REPEATS else "-", o["partials"], o["lnl"], o["sumtable"], o["derivatives"],
            o["lnl_us"], o["deriv_us"], ("<-- " + ",".join(flags)) if flags else ""), flush=True)


if __name__ == "__main__":
    main()
