#!/usr/bin/env python3
"""developer tool: the last N dispatches of a rocprofv3 --kernel-trace directory as a timeline
   (start relative to the first shown, duration, gap to the one before), names shortened.
   python3 tools/kernel_timeline.py DIR [N]"""
import csv, glob, os, re, sys
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 80
for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))[-last:]
    t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        m = re.search(r"(k_\w+|[A-Za-z_]*(Sort|Scan|Histogram|Onesweep|onesweep|scan|sort|histogram)\w*|__amd_\w+)", r["Kernel_Name"])
        print("%10.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, (m.group(0) if m else r["Kernel_Name"])[:70]))
        prev_end = e
