#!/usr/bin/env python3
"""Runs ON THE GPU BOX: one contig with more than 2^32 bases (base offsets beyond 32 bits) through the
engine; the tail of the output is compared bit for bit with the CPU oracle run on the tail reads alone
(preset -9: no running-depth state, so columns a few kb inside the tail see the same pileup).
Usage: python tools/large_offset_check.py [scale=2.4]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import crumble_amd as ca
from crumble_amd import HostBatch
import synth
import oracle_lib as O

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.4
t0 = time.time()
hb = synth.generate(synth.config("chr20", scale=scale))
print(f"generated {hb.n_reads} reads, {hb.n_bases} bases ({hb.n_bases / 2**32:.2f} x 2^32) in {time.time() - t0:.1f} s", flush=True)
assert hb.n_bases > 2**32
p = ca.preset_params(9)
eng = ca.Engine(p, 0)
t0 = time.time()
eng.upload(hb); t_up = time.time() - t0
eng.run(); eng.run()
ms = sum(m for _, m, _ in eng.kernel_times())
q, f, s = eng.download()
print(f"upload {t_up:.2f} s, kernels {ms:.1f} ms = {hb.n_bases / ms / 1e6:.1f} Gbases/s", flush=True)
# tail: the last ~800k reads as their own batch
n_tail = 800_000
r0 = hb.n_reads - n_tail
b0, c0 = int(hb.seq_off[r0]), int(hb.cig_off[r0])
assert b0 > 2**32, "tail must lie beyond the 32-bit boundary"
tail = HostBatch(hb.tid, hb.pos[r0:], hb.mapq[r0:], hb.flag[r0:], hb.seq_off[r0:] - b0, hb.cig_off[r0:] - c0,
                 hb.cigar[c0:], hb.bases[b0:], hb.quals[b0:])
O.load()
wq, wf, ws = O.contig(p, tail)
inner = np.nonzero(tail.pos > tail.pos[0] + 20000)[0]       # reads whose every column is fully inside the tail
lo = int(tail.seq_off[inner[0]])
same = np.array_equal(q[b0 + lo:], wq[lo:])
print("tail reads compared:", len(inner), "bases:", len(wq) - lo, "bit-exact:", same, flush=True)
assert same and np.array_equal(f[r0 + inner[0]:], wf[inner[0]:])
# whole-array properties
nb_stat = int(s[ca.capi.ST_NAMES.index("bases")])
assert nb_stat % hb.n_bases == 0 and int(s[32:].sum()) == nb_stat, (nb_stat, hb.n_bases, int(s[32:].sum()))
print("OK")
