import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import crumble_amd as ca
import synth
eng = ca.Engine(ca.preset_params(5), 0)
cfg = synth.config("chr20", scale=1.0); cfg.contig_len = 50000
hb = synth.generate(cfg)
eng.upload(hb)
for _ in range(50):
    eng.run()
