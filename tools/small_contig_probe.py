#!/usr/bin/env python3
"""GPU box: cost of crumble_hip_contig() on small contigs (alt / decoy / unplaced sequences of a real
reference: thousands of them, a few kb to a few hundred kb each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import crumble_amd as ca
import synth
eng = ca.Engine(ca.preset_params(5), 0)
sizes = [int(a) for a in sys.argv[1:]] or [5000, 50000, 500000, 5000000]
for clen in sizes:
    cfg = synth.config("chr20", scale=1.0); cfg.contig_len = clen
    hb = synth.generate(cfg)
    eng.contig(hb)
    n = 200 if clen <= 500000 else 20
    t0 = time.perf_counter()
    for _ in range(n):
        eng.contig(hb)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run()
    dr = (time.perf_counter() - t0) / n
    print(f"contig {clen} bp, {hb.n_reads} reads, {hb.n_bases} bases: contig() {dt*1e3:.3f} ms, run() alone {dr*1e3:.3f} ms", flush=True)
