#!/usr/bin/env python3
"""Prints the last N dispatches of a rocprofv3 --kernel-trace CSV as a timeline (start offset, duration, kernel, grid).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python tools/probe_counters.py ; python tools/kernel_timeline.py /tmp/p 40
"""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    t0 = s if t0 is None else t0
    print("%9.1f us  +%8.1f us  %-62s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:62], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
