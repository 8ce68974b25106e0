#!/usr/bin/env python3
"""developer tool: the per-kernel lines of a rocprofv3 --kernel-trace --stats directory, names shortened.
   python3 tools/kernel_stats.py DIR [top]"""
import csv, glob, os, re, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for path in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
    for row in list(csv.DictReader(open(path)))[:top]:
        m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", row["Name"])
        print("%-44s calls %6s  avg %9.1f ns  min %8s  max %8s" % ((m.group(0) if m else row["Name"])[:44], row["Calls"], float(row["AverageNs"]), row["MinNs"], row["MaxNs"]))
