#!/usr/bin/env python3
"""Largest launches per kernel of a rocprofv3 kernel trace (the full-size launches of a run that also makes small ones).
Usage: tools/ktop.py <dir with *_kernel_trace.csv> [top n]"""
import collections
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:64]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, v in d.items():
    v.sort(reverse=True)
    t = v[:top]
    rows.append((sum(t) / len(t), k, len(v)))
for a, k, n in sorted(rows, reverse=True)[:30]:
    print("%9.1f us (top%d avg) n=%4d %s" % (a, top, n, k))
