#!/usr/bin/env python3
"""Runs ON THE GPU BOX: randomised stress of the device BGZF codec against zlib.  Every case builds a byte string out of
random pieces (noise, runs, periodic patterns, BAM-like records, repeats of earlier material at random distances, zeros),
then checks  zlib-inflate(device-deflate(x)) == x  (gzip members, CRC-32 and ISIZE checked by zlib),
device-inflate(zlib-deflate(x)) == x at a random level / strategy / block size, with both token kernels, and
device-inflate(device-deflate(x)) == x.  Usage: tools/gz_fuzz.py [seed] [seconds]"""
import os
import struct
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from crumble_amd import gz
import test_gz as T

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
c = gz.GzCodec(0)


def piece(n, so_far):
    k = int(rng.integers(0, 8))
    if k == 0:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if k == 1:
        return bytes([int(rng.integers(0, 256))]) * n
    if k == 2:
        p = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
        return (p * (n // len(p) + 1))[:n]
    if k == 3:
        return T.samples(rng, n)["records"]
    if k == 4 and so_far:                                   # material from further back (up to and beyond 32 KiB)
        d = int(rng.integers(1, min(len(so_far), 70000) + 1))
        src = so_far[-d:]
        return (src * (n // len(src) + 1))[:n]
    if k == 5:
        return T.samples(rng, n)["bases"]
    if k == 6:
        return T.samples(rng, n)["runs"]
    return bytes(n)


t0 = time.time()
cases = 0
total = 0
while time.time() - t0 < budget:
    size = int(rng.choice([0, 1, 2, 3, 100, 5000, 65279, 65280, 65281, 200000, 1500000, int(rng.integers(0, 3000000))]))
    data = b""
    while len(data) < size:
        data += piece(int(min(size - len(data), rng.integers(1, 90000))), data)
    data = data[:size]
    # device deflate -> zlib
    for level in (6, 0) if cases % 5 == 0 else (6,):
        comp = c.deflate(data, level).tobytes()
        back, nm = T.gunzip_members(comp)
        assert back == data, ("deflate", seed, cases, size, level)
        assert c.inflate(comp).tobytes() == data, ("own", seed, cases, size)
    # zlib -> device inflate
    level = int(rng.choice([0, 1, 4, 6, 9]))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
    blk = int(rng.choice([0xff00, 0xff00, 1000, 30000, 65000]))
    if level == 0:
        blk = min(blk, 65000)
    try:
        zc = T.bgzf(data, level, blk=blk, strategy=strategy)
    except AssertionError:                                  # (a member that zlib could not fit into 64 KiB: noise at a large block size)
        zc = T.bgzf(data, level, blk=30000, strategy=strategy)
    for tokens, lpw in (("wave", "auto"), ("lane", str(rng.choice([8, 16, 64])))):
        c.set_option("tokens", tokens)
        c.set_option("lanes_per_wave", lpw)
        assert c.inflate(zc).tobytes() == data, ("inflate", seed, cases, size, level, strategy, blk, tokens, lpw)
    c.set_option("tokens", "auto")
    c.set_option("lanes_per_wave", "auto")
    cases += 1
    total += size
c.close()
print(f"gz_fuzz seed {seed}: {cases} cases, {total / 1e6:.1f} MB, 0 failures, {time.time() - t0:.0f} s")
