#!/usr/bin/env python3
"""Runs ON THE GPU BOX: a file of many small contigs (hs38DH's decoys and alts: thousands of a few kb) through crumble-hip's
device and host pipelines: the loop times, and the device pipeline with 1-4 engines.  Usage: tools/small_contigs.py [n] [bp]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from crumble_amd import bamio

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
bp = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
tmp = os.environ.get("TMPDIR", "/tmp")
src, dst = os.path.join(tmp, "sc_in.bam"), os.path.join(tmp, "sc_out.bam")
import copy
batches, refs, pool = [], [], []
for k in range(8):                                              # eight different contigs, taken in turn (generating thousands takes minutes)
    cfg = synth.config("chr20", scale=1.0)
    cfg.contig_len = bp
    cfg.seed = 9000 + k
    pool.append(synth.generate(cfg))
for tid in range(n):
    b = copy.copy(pool[tid % 8])
    b.tid = tid
    batches.append(b)
    refs.append((f"d{tid}", bp + 1000))
bamio.write_batches(src, refs, batches, level=1)
total = sum(b.n_bases for b in batches)
print(f"{n} contigs of {bp} bp: {total / 1e6:.1f} Mbases, {os.path.getsize(src) / 1e6:.1f} MB", flush=True)
for name, env in (("host, 2 engines", {"CRUMBLE_HIP_PIPELINE": "host"}), ("host, 4 engines", {"CRUMBLE_HIP_PIPELINE": "host", "CRUMBLE_HIP_ENGINES": "4"}),
                  ("device, 1 engine", {"CRUMBLE_HIP_ENGINES": "1"}), ("device, 2", {"CRUMBLE_HIP_ENGINES": "2"}), ("device, 3", {"CRUMBLE_HIP_ENGINES": "3"}), ("device, 4 (default)", {})):
    best = None
    for rep in range(2):
        r = subprocess.run([os.path.join(ROOT, "crumble_amd", "crumble-hip"), "-v", "-5", "-z", src, dst], env=dict(os.environ, CRUMBLE_BAM_LEVEL="1", **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode:
            print(name, "FAILED", r.stderr[-300:])
            break
        if ("pipeline: device" in r.stderr) != name.startswith("device"):
            print(name, "ran the other pipeline:", [l for l in r.stderr.splitlines() if "pipeline" in l][:2])
        t = float(re.search(r"end-to-end ([0-9.]+) s", r.stderr).group(1))
        best = t if best is None else min(best, t)
    if best:
        print(f"{name:22s} loop {best:.3f} s = {best / n * 1e3:.3f} ms per contig, {total / best / 1e9:.3f} Gbases/s", flush=True)
os.remove(src); os.remove(dst)
