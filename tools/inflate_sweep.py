#!/usr/bin/env python3
"""Runs ON THE GPU BOX: kernel time of the device inflate (k_gz_tokens / k_gz_tokens_lanes + k_gz_resolve) on BAMs of several
sizes, for every walk: a wave per member, a lane per member with 8 / 16 / 64 walking lanes per wave.
Usage: tools/inflate_sweep.py [scale ...]   (scale 0.3 = a gigabyte of records)"""
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from crumble_amd import bamio, gz
import synth

scales = [float(x) for x in sys.argv[1:]] or [0.02, 0.2]
c = gz.GzCodec(0)
for scale in scales:
    hb = synth.generate(synth.config("chr20", scale=scale))
    path = "/tmp/gz_sweep.bam"
    bamio.write_batches(path, [("chr20", 70000000)], [hb], level=1)
    comp = np.fromfile(path, np.uint8)
    nblk = len(gz.bgzf_scan(comp)[0])
    want = None
    for tokens, lpw in (("wave", "auto"), ("lane", "8"), ("lane", "16"), ("lane", "64"), ("auto", "auto")):
        c.set_option("tokens", tokens)
        c.set_option("lanes_per_wave", lpw)
        best = None
        for rep in range(3):
            got = c.inflate(comp)
            tm = c.times()
            best = tm["kernels_ms"] if best is None else min(best, tm["kernels_ms"])
        if want is None:
            want = got.copy()
        assert np.array_equal(got, want)
        print(f"scale {scale}: {nblk} members, {got.size / 1e6:.0f} MB raw: {tokens:5s} lpw {lpw:4s} kernels {best:8.2f} ms = {got.size / best / 1e6:7.1f} GB/s", flush=True)
    assert want.tobytes() == gzip.decompress(comp.tobytes())
c.close()
