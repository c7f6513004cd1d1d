#!/usr/bin/env python3
"""Summarises hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin: the make output) as one line per kernel."""
import re
import subprocess
import sys

cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
except Exception:
    dem = names
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    print(f"{d:48s} vgpr {r.get('VGPRs', '?'):>4} sgpr {r.get('TotalSGPRs', '?'):>4} occ {r.get('Occupancy [waves/SIMD]', '?'):>2} "
          f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>5} spill {r.get('VGPRs Spill', '?'):>3} lds {r.get('LDS Size [bytes/block]', '?'):>7}")
