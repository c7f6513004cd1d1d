#!/usr/bin/env python3
"""Runs ON THE GPU BOX: kernel time of the device inflate with parts of k_gz_resolve left out (diagnostic builds:
tools/build_variant.sh nojump -DGZ_EXP_NOJUMP, notokw -DGZ_EXP_NOTOKW; wrong bytes, so the CRC check is off for those) -- what a
part costs is what the kernel gains without it.  Usage: tools/resolve_variants.py [scale]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import crumble_amd as ca
from crumble_amd import bamio, gz
import synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
hb = synth.generate(synth.config("chr20", scale=scale))
bamio.write_batches("/tmp/rv.bam", [("chr20", 70000000)], [hb], level=1)
comp = np.fromfile("/tmp/rv.bam", np.uint8)
for tag, crc in (("", True), ("", False), ("nojump", False), ("notokw", False)):
    L = ca.load_library(os.path.join(ROOT, "build", f"libcrumble_{tag}.so"), make_default=False) if tag else None
    c = gz.GzCodec(0, L)
    best = None
    for rep in range(3):
        c.inflate(comp, check_crc=crc)
        t = c.times()["kernels_ms"]
        best = t if best is None else min(best, t)
    print(f"variant [{tag or 'whole'}] crc {crc}: kernels {best:.2f} ms", flush=True)
    c.close()
