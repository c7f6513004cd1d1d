#!/usr/bin/env python3
"""Runs ON THE GPU BOX with CRUMBLE_HIP_LIB=build/libcrumble_gzstamp.so (tools/build_variant.sh gzstamp -DGZ_STAMP):
shader cycles per phase of k_gz_deflate on a BAM-shaped stream."""
import ctypes as C, gzip, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import crumble_amd as ca
from crumble_amd import bamio, gz
import synth
L = ca.load_library(os.environ["CRUMBLE_HIP_LIB"])
hb = synth.generate(synth.config("chr20", scale=0.05))
bamio.write_batches("/tmp/s.bam", [("chr20", 70000000)], [hb], level=1)
want = np.frombuffer(gzip.decompress(open("/tmp/s.bam", "rb").read()), np.uint8)
c = gz.GzCodec(0, L)
out = (C.c_ulonglong * 16)()
L.crumble_gz_debug_stamps(out, 1)
c.deflate(want, 6)
L.crumble_gz_debug_stamps(out, 1)
names = ["load+crc+eq", "phase1 candidates", "phase2 parse/insert", "phase3 tokens", "huffman", "header", "encode+store", "-"]
tot = sum(out[i] for i in range(8))
nb = (want.size + 65279) // 65280
for i in range(7):
    print(f"{names[i]:22s} {out[i] / nb:12.0f} cycles/block  {100.0 * out[i] / tot:5.1f} %")
for i, nm in ((8, "pass B: candidates loaded"), (9, "pass B: walk"), (10, "pass B: tokens, histogram")):
    print(f"{nm:28s} {out[i] / nb:12.0f} cycles/block (wave 0 of 16)")
print("blocks", nb, "cycles/block", tot / nb, c.times())
c.close()
