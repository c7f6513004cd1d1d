#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the device BGZF codec (crumble_gz_*) on a BAM written by the host writer -- kernel rates of inflate
and deflate, the PCIe-inclusive rates, sizes against the host codec's.  Usage: tools/gz_probe.py [scale] [level]"""
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import crumble_amd as ca
from crumble_amd import bamio, gz
import synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
hb = synth.generate(synth.config("chr20", scale=scale))
path = "/tmp/gz_probe.bam"
t0 = time.time()
bamio.write_batches(path, [("chr20", 70000000)], [hb], level=level)
t_host_deflate = time.time() - t0
comp = np.fromfile(path, np.uint8)
t0 = time.time()
want = np.frombuffer(gzip.decompress(comp.tobytes()), np.uint8)
t_py_inflate = time.time() - t0
c = gz.GzCodec(0)
if os.environ.get('GZ_TOKENS'):
    c.set_option('tokens', os.environ['GZ_TOKENS'])
if os.environ.get('GZ_LPW'):
    c.set_option('lanes_per_wave', os.environ['GZ_LPW'])
out = {"bases": int(hb.n_bases), "raw_bytes": int(want.size), "comp_bytes_host": int(comp.size), "host_codec": bamio.codec() if hasattr(bamio, "codec") else "?",
       "host_write_s": round(t_host_deflate, 3), "python_gunzip_s": round(t_py_inflate, 3)}
for rep in range(3):
    t0 = time.time()
    got = c.inflate(comp)
    wall = time.time() - t0
    tm = c.times()
    if rep == 0:
        assert np.array_equal(got, want)
    out[f"inflate_{rep}"] = {"wall_s": round(wall, 4), **{k: round(v, 3) for k, v in tm.items()},
                             "kernel_GBps_out": round(want.size / tm["kernels_ms"] / 1e6, 2), "wall_GBps_out": round(want.size / wall / 1e9, 2)}
for rep in range(3):
    t0 = time.time()
    comp2 = c.deflate(want, 6)
    wall = time.time() - t0
    tm = c.times()
    if rep == 0 and not os.environ.get('GZ_PROBE_NOCHECK'):
        assert gzip.decompress(comp2.tobytes()) == want.tobytes()
    out[f"deflate_{rep}"] = {"wall_s": round(wall, 4), **{k: round(v, 3) for k, v in tm.items()}, "comp_bytes": int(comp2.size),
                             "ratio_vs_host": round(comp2.size / comp.size, 4),
                             "kernel_GBps_in": round(want.size / tm["kernels_ms"] / 1e6, 2), "wall_GBps_in": round(want.size / wall / 1e9, 2)}
c.close()
print(json.dumps(out))
