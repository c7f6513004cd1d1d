"""GPU box: sizes of the run's work lists (trigger columns, triggers) relative to the batch, per configuration."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import crumble_amd as ca, synth
for wl, sc, preset, tech in (("chr20", 0.05, 5, None), ("chr20", 0.05, 9, None), ("panel", 0.2, 9, None), ("hifi", 0.02, None, "pbccs")):
    hb = synth.generate(synth.config(wl, scale=sc))
    eng = ca.Engine(ca.preset_params(preset, tech=tech), 0)
    q, f, s = eng.contig(hb)
    cols = eng.columns()
    L = int(cols["pos"].max() - cols["pos"].min() + 1)
    ntrig = int((cols["trigger"] != 0).sum())
    print(wl, preset, "L", L, "nb", hb.n_bases, "reads", hb.n_reads, "trigcols", ntrig, "items", int(s[ca.ST_NAMES.index("triggers")]),
          "trig/L", ntrig / L, "items/nb", s[ca.ST_NAMES.index("triggers")] / hb.n_bases, "items/trig", s[ca.ST_NAMES.index("triggers")] / max(1, ntrig))
    eng.close()
