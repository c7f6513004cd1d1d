"""GPU box: P0 (device-side expansion) time of a first and of a repeated upload of the same batch."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import crumble_amd as ca
import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "chr20"
hb = synth.generate(synth.config(wl, scale=float(sys.argv[2]) if len(sys.argv) > 2 else 1.0))
e = ca.Engine(ca.preset_params(5), 0)
for i in range(3):
    t0 = time.perf_counter(); e.upload(hb); t = time.perf_counter() - t0
    print(f"upload {i}: wall {t*1e3:.1f} ms, h2d {e.upload_times()[0]:.1f} ms, p0 {e.upload_times()[1]:.2f} ms")
