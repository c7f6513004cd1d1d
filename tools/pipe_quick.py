#!/usr/bin/env python3
"""Runs ON THE GPU BOX: one WGS-shaped file (24 contigs, bench.py's end_to_end workload x scale) through crumble-hip under
several environments, the driver's own end-to-end line for each.  Usage: tools/pipe_quick.py [scale] [reps] [env-set ...]
where an env-set is  name:K=V,K=V  (default: the host pipeline and the device pipeline at level 1 and the default level)."""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from crumble_amd import bamio, shard

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sets = sys.argv[3:] or ["host1:CRUMBLE_HIP_PIPELINE=host,CRUMBLE_BAM_LEVEL=1", "dev1:CRUMBLE_HIP_PIPELINE=device,CRUMBLE_BAM_LEVEL=1",
                        "host6:CRUMBLE_HIP_PIPELINE=host", "dev6:CRUMBLE_HIP_PIPELINE=device"]
tmp = os.environ.get("TMPDIR", "/tmp")
src, dst = os.path.join(tmp, "pq_in.bam"), os.path.join(tmp, "pq_out.bam")
sizes = [int(l * 0.005 * scale) for l in shard.GRCH38]
batches = []
for tid, ln in enumerate(sizes):
    cfg = synth.config("chr20", scale=1.0)
    cfg.contig_len = max(ln, 4 * cfg.read_len + 4096)
    cfg.seed = 7000 + tid
    cfg.tid = tid
    batches.append(synth.generate(cfg))
refs = [(f"chr{t + 1}", max(s_, 4 * 150 + 4096) + 1000) for t, s_ in enumerate(sizes)]
bamio.write_batches(src, refs, batches, level=1)
total = sum(b.n_bases for b in batches)
del batches
print(f"file: {os.path.getsize(src) / 1e6:.1f} MB, {total / 1e6:.1f} Mbases", flush=True)
ref_stream = None
for rep in range(reps):
    for s in sets:
        name, _, kv = s.partition(":")
        env = dict(os.environ)
        for x in kv.split(","):
            if x:
                k, _, v = x.partition("=")
                env[k] = v
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "crumble_amd", "crumble-hip"), "-v", "-5", "-z", src, dst], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if r.returncode:
            print(name, "FAILED", r.stderr[-400:], flush=True)
            continue
        m = re.search(r"end-to-end ([0-9.]+) s \(read\+parse ([0-9.]+), upload\+kernels\+download ([0-9.]+), write ([0-9.]+)\)", r.stderr)
        st = [float(x) for x in m.groups()]
        extra = (re.search(r"pipeline: device \((.*)\)", r.stderr) or [None, ""])[1]
        print(f"{name:8s} rep {rep}: {total / st[0] / 1e9:.3f} Gbases/s  loop {st[0]:.3f} read {st[1]:.3f} gpu {st[2]:.3f} write {st[3]:.3f}  process {wall:.2f} s  out {os.path.getsize(dst) / 1e6:.1f} MB  {extra[-110:]}", flush=True)
        if rep == 0:
            import gzip
            import hashlib
            h = hashlib.sha256(gzip.decompress(open(dst, "rb").read())).hexdigest()[:16]
            if ref_stream is None:
                ref_stream = h
            print(f"         inflated output sha {h} {'== first' if h == ref_stream else '!= FIRST'}", flush=True)
if not os.environ.get('PQ_KEEP'):
    os.remove(src)
os.remove(dst)
