#!/usr/bin/env python3
"""Where a 5 kb contig's fixed cost goes: wall time of crumble_hip_contig() and of its parts, per call (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import crumble_amd as ca
import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = synth.config("chr20", scale=1.0)
cfg.contig_len = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
hb = synth.generate(cfg)
hb.pack4(ca.lib()).lens()
eng = ca.Engine(ca.preset_params(5), 0)
for _ in range(10):
    eng.contig(hb)

def t(f, n=reps):
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3

print("contig      %.4f ms" % t(lambda: eng.contig(hb)))
print("upload      %.4f ms" % t(lambda: eng.upload(hb)))
print("upload_raw  %.4f ms" % t(lambda: eng.upload_raw(hb)))
eng.upload(hb)
print("expand      %.4f ms" % t(lambda: eng.expand()))
print("run         %.4f ms" % t(lambda: eng.run()))
eng.run()
print("download    %.4f ms" % t(lambda: eng.download(want_stats=True)))
print("encode+dl   %.4f ms" % t(lambda: eng.download_coded()))
eng.close()
