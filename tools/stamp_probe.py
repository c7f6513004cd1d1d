import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, crumble_amd as ca
import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "chr20"
cfg = synth.config(wl, scale=float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
lvl = sys.argv[3] if len(sys.argv) > 3 else "5"
if lvl in ("pbccs", "pbccs_nobed"):          # config 4: -y pbccs -R keep.bed (pbccs_nobed: without the -R)
    import bench
    hb = synth.generate(cfg, bed=bench.keep_bed(cfg.contig_len, int(5000 * cfg.contig_len / 248956422 + 0.5)) if lvl == "pbccs" else None)
    e = ca.Engine(ca.preset_params(None, tech="pbccs"), 0)
else:
    hb = synth.generate(cfg)
    e = ca.Engine(ca.preset_params(int(lvl)), 0)
e.upload(hb); e.run(); e.stats_reset(); e.run()
s = e.stats_read()
kt = dict((n, ms) for n, ms, _ in e.kernel_times())
t = s[24:30].astype(float)
print("k_columns2 ms", kt.get("k_columns2"), "stamps (share): unit-grab %.3f stage %.3f produce %.3f consume(rows+issue) %.3f decide %.3f consume(LDS wait) %.3f" % tuple(t / t.sum()), "total wave-cycles %.3g" % t.sum())
print("passes %d row-steps %d (%.1f per pass); lane-steps / bases = %.2f; wave-cycles per step in consume %.0f" % (s[30], s[31], s[31] / max(1, s[30]), s[31] * 64 / hb.n_bases, (t[3] + t[5]) / max(1, s[31])))
