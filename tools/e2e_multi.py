"""GPU box: file -> file through crumble-hip on a multi-contig BAM (contig pipeline: read / GPU / write overlap)."""
import os, sys, time, subprocess, re
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from crumble_amd import bamio
import synth
ROOT = os.getcwd()
n, scale = int(sys.argv[1]) if len(sys.argv) > 1 else 6, float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
tmp = os.environ.get("TMPDIR", "/tmp")
src, dst = os.path.join(tmp, "multi_in.bam"), os.path.join(tmp, "multi_out.bam")
bs, refs = [], []
for t in range(n):
    cfg = synth.config("chr20", scale=scale); cfg.seed = 900 + t; cfg.tid = t
    bs.append(synth.generate(cfg)); refs.append((f"c{t}", cfg.contig_len))
bamio.write_batches(src, refs, bs, level=1)
bases = sum(b.n_bases for b in bs)
for j in sys.argv[3:] or ["1"]:
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "crumble_amd", "crumble-hip"), "-v", "-5", "-j", j, src, dst],
                       env=dict(os.environ, CRUMBLE_BAM_LEVEL="1", CRUMBLE_HIP_DEVICES=",".join("0" * int(j))), stderr=subprocess.PIPE, text=True)
    wall = time.perf_counter() - t0
    m = re.search(r"end-to-end.*", r.stderr)
    print(f"-j {j}: process wall {wall:.2f} s = {bases / wall / 1e9:.3f} Gbases/s; {m.group(0) if m else r.stderr[-300:]}")
os.remove(src); os.remove(dst)
