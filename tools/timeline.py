#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace csv directory: events longer than a threshold and the idle
gaps between consecutive events.  Usage: tools/timeline.py <dir> [min_event_ms] [min_gap_ms]"""
import csv
import glob
import sys

d = sys.argv[1]
min_ev = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"][12:]))
ev.sort()
t0 = last = ev[0][0]
busy = 0
for s, e, n in ev:
    if (s - last) / 1e6 > min_gap:
        print(f"          --- idle {(s - last) / 1e6:.2f} ms (before {n})")
    if (e - s) / 1e6 > min_ev:
        print(f"{(s - t0) / 1e6:9.2f} +{(e - s) / 1e6:7.2f} ms {n}")
    busy += max(0, e - max(s, last))
    last = max(last, e)
print(f"span {(last - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms")
