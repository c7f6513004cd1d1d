"""Randomised (options x synthetic input) cases for the parity fuzz tests (CPU emulator and GPU)."""
import numpy as np

import crumble_amd as ca
import synth


def random_case(seed, max_bases=250_000):
    """-> (params, HostBatch, description).  Every option of the engine is drawn from the ranges the
    reference's command line accepts (fenced: qualities / caps <= 100)."""
    rng = np.random.default_rng(seed)
    name = rng.choice(["chr20", "ce", "panel", "hifi"], p=[0.4, 0.25, 0.2, 0.15])
    over = dict(seed=int(rng.integers(1, 1 << 30)))
    if name == "chr20":
        scale = 0.0002 * (1 + rng.random())
        over.update(depth=float(rng.choice([3, 12, 30, 60])), read_len=int(rng.choice([36, 100, 150, 251])))
    elif name == "ce":
        scale = 0.3 + rng.random()
    elif name == "panel":
        scale = 0.0008
        over.update(depth=float(rng.choice([40, 150, 300])), spike_depth=int(rng.choice([0, 1500])))
    else:
        scale = 0.0002
        over.update(read_len=int(rng.choice([1200, 3000, 6000])), depth=float(rng.choice([8, 20])))
    over.update(err_indel=float(rng.choice([1e-4, 1e-3, 5e-3])), refskip_frac=float(rng.choice([0, 0.01, 0.05])),
                softclip_frac=float(rng.choice([0.0, 0.05, 0.3])), mapq0_frac=float(rng.choice([0.0, 0.05, 0.4])),
                unmapped_frac=float(rng.choice([0.0, 0.01])))
    cfg = synth.config(name, scale=scale, **over)
    hb = synth.generate(cfg)
    if hb.n_bases > max_bases:                       # keep the oracle + emulator fast
        from bench import prefix_batch
        n = int(np.searchsorted(hb.seq_off, max_bases))
        hb = prefix_batch(hb, max(n, 1))
    if rng.random() < 0.4 and hb.n_reads:
        lo, hi = int(hb.pos.min()), int(hb.pos.max()) + 200
        st = np.sort(rng.integers(lo, hi, int(rng.integers(1, 12)))).astype(np.int32)
        bed = np.stack([st, st + rng.integers(1, 400, len(st)).astype(np.int32)], 1)
        t = np.zeros(len(bed), np.int32)
        s_, e_ = bed[:, 0].copy(), bed[:, 1].copy()
        n = ca.lib().crumble_bed_collapse(t.ctypes.data, s_.ctypes.data, e_.ctypes.data, len(bed))
        hb.bed_start, hb.bed_end = s_[:n].copy(), e_[:n].copy()
    level = int(rng.choice([1, 3, 5, 7, 8, 9]))
    kw = {}
    if rng.random() < 0.3: kw["tech"] = "pbccs"
    if rng.random() < 0.3: kw["keep"] = str(rng.choice(["2", "11,25", "30-35", "93"]))
    if rng.random() < 0.2: kw["really_keep"] = str(rng.choice(["37", "2-5", "40"]))
    o = {}
    if rng.random() < 0.3: o["min_qual_A"] = int(rng.choice([0, 20, 60]))
    if rng.random() < 0.3: o["min_qual_B"] = int(rng.choice([0, 40, 75, 100]))
    if rng.random() < 0.3: o["min_indel_A"] = int(rng.choice([0, 50, 100]))
    if rng.random() < 0.3: o["min_indel_B"] = int(rng.choice([0, 100, 200]))
    if rng.random() < 0.3: o["min_discrep_A"] = float(rng.choice([0.0, 1.0, 2.5]))
    if rng.random() < 0.3: o["min_discrep_B"] = float(rng.choice([0.0, 0.8, 1.5]))
    if rng.random() < 0.3: o["min_mqual"] = int(rng.choice([0, 5, 30]))
    if rng.random() < 0.3: o["qlow"] = int(rng.choice([2, 5, 10]))
    if rng.random() < 0.3: o["qhigh"] = int(rng.choice([30, 40, 60]))
    if rng.random() < 0.3: o["qcutoff"] = int(rng.choice([10, 25, 40]))
    if rng.random() < 0.3: o["qcap"] = int(rng.choice([40, 60, 100]))
    if rng.random() < 0.3: o["pblock"] = int(rng.choice([0, 1, 4, 8, 16]))
    if rng.random() < 0.2: o["softclip"] = 1
    if rng.random() < 0.2: o["perfect_col"] = 1
    if rng.random() < 0.2: o["binary_qual"] = 1
    if rng.random() < 0.1: o["reduce_qual"] = 0
    if rng.random() < 0.3: o["indel_fract"] = float(rng.choice([0.0, 0.1, 0.5]))
    if rng.random() < 0.3: o["low_mqual_perc"] = float(rng.choice([0.0, 0.1, 0.5, 1.0]))
    if rng.random() < 0.3: o["clip_perc"] = float(rng.choice([0.0, 0.05, 0.2, 1.0]))
    if rng.random() < 0.3: o["ins_len_perc"] = float(rng.choice([0.0, 0.1, 1.0]))
    if rng.random() < 0.3: o["over_depth"] = float(rng.choice([0.5, 1.2, 3.0, 999.0]))
    if rng.random() < 0.3: o["indel_ov_perc"] = float(rng.choice([0.0, 0.2, 0.5]))
    if rng.random() < 0.3: o["iSTR_mul"] = float(rng.choice([0.0, 1.0, 2.0])); o["iSTR_add"] = int(rng.choice([0, 2, 5]))
    if rng.random() < 0.3: o["sSTR_mul"] = float(rng.choice([0.0, 1.0, 2.0])); o["sSTR_add"] = int(rng.choice([0, 2, 5]))
    p = ca.preset_params(level, **kw, **o)
    return p, hb, f"seed {seed}: {name} {over} -{level} {kw} {o} bed={len(hb.bed_start)}"
