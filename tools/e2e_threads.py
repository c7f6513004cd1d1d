#!/usr/bin/env python3
"""GPU box: crumble-hip file -> file on the 24-contig WGS-shaped file with different host-thread settings."""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from crumble_amd import bamio, shard
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.005
sizes = [int(l * scale) for l in shard.GRCH38]
batches = []
for tid, ln in enumerate(sizes):
    cfg = synth.config("chr20", scale=1.0); cfg.contig_len = max(ln, 4700); cfg.seed = 7000 + tid; cfg.tid = tid
    batches.append(synth.generate(cfg))
refs = [(f"chr{t + 1}", max(s_, 4700) + 1000) for t, s_ in enumerate(sizes)]
tmp = tempfile.mkdtemp(prefix="e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
src, dst = os.path.join(tmp, "in.bam"), os.path.join(tmp, "out.bam")
bamio.write_batches(src, refs, batches, level=1)
total = sum(b.n_bases for b in batches)
for label, env_extra, args in (("default", {}, []), ("-@ 8", {}, ["-@", "8"]), ("-@ 6", {}, ["-@", "6"]), ("-@ 10", {}, ["-@", "10"]), ("-@ 12", {}, ["-@", "12"])):
    env = dict(os.environ, CRUMBLE_BAM_LEVEL="1", **env_extra)
    best = None
    for _ in range(2):
        r = subprocess.run([os.path.join(ROOT, "crumble_amd", "crumble-hip"), "-v", "-5"] + args + [src, dst], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        m = re.search(r"end-to-end ([0-9.]+) s \(read\+parse ([0-9.]+), upload\+kernels\+download ([0-9.]+), write ([0-9.]+)\)", r.stderr)
        if r.returncode or m is None:
            print(f"{label:16s} crumble-hip failed ({r.returncode}): {r.stderr[-300:]}", flush=True)
            continue
        st = [float(x) for x in m.groups()]
        if best is None or st[0] < best[0]: best = st
    if best is None:
        continue
    print(f"{label:16s} loop {best[0]:.3f} s = {total / best[0] / 1e9:.2f} Gbases/s (read {best[1]:.3f} gpu {best[2]:.3f} write {best[3]:.3f})", flush=True)
    bl = [l for l in r.stderr.splitlines() if l.startswith("bamio:")]
    if bl: print("   ", bl[-1])
os.remove(src); os.remove(dst); os.rmdir(tmp)
