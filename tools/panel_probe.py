#!/usr/bin/env python3
"""GPU box: k_columns time on the panel workload with and without the 6000x spike, and at other depths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crumble_amd as ca
import synth
for kw in (dict(), dict(spike_depth=0), dict(spike_depth=0, depth=100.0), dict(spike_depth=0, depth_sigma=0.0)):
    hb = synth.generate(synth.config("panel", scale=1.0, **kw))
    eng = ca.Engine(ca.preset_params(9), 0)
    eng.upload(hb)
    eng.run(); eng.run()
    t = {n: m for n, m, _ in eng.kernel_times()}
    print(os.environ.get("CRUMBLE_HIP_LIB", "default")[-22:], kw, hb.n_bases, "k_columns %.3f ms" % t["k_columns"], flush=True)
    eng.close()
