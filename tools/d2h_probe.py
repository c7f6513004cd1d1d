"""GPU box: crumble_hip_download rate (device -> pageable host memory through the staging lanes) for several chunk sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import crumble_amd as ca
import synth
hb = synth.generate(synth.config("chr20", scale=float(sys.argv[1]) if len(sys.argv) > 1 else 0.5))
for chunk in (None, 1 << 20, 2 << 20, 4 << 20):
    e = ca.Engine(ca.preset_params(5), 0, options={"dl_chunk": str(chunk)} if chunk else None)
    e.upload(hb); e.run()
    q, f, _ = e.download(want_stats=False)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); e.download(want_stats=False, out=(q, f)); ts.append(time.perf_counter() - t0)
    print(f"chunk {chunk or 8 << 20:>9}: {hb.n_bases / min(ts) / 1e9:6.1f} GB/s  ({min(ts) * 1e3:.1f} ms for {hb.n_bases / 1e9:.2f} GB)")
    e.close()
