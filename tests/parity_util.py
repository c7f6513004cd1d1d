"""Shared helpers for the parity tests (HIP engine or emulated engine vs oracle)."""
import numpy as np

import crumble_amd as ca
from crumble_amd import HostBatch
import synth
import oracle_lib as O

_cache = {}


def synth_batch(name, scale, **over):
    key = (name, scale, tuple(sorted(over.items())))
    if key not in _cache:
        _cache[key] = synth.generate(synth.config(name, scale=scale, **over))
    return _cache[key]


def assert_parity(make_engine, params, hb, check_columns=True):
    """Runs crumble_hip_contig on `hb` and requires bit-exact equality with the oracle."""
    want = O.contig(params, hb, trace=check_columns)
    q, f, s = want[:3]
    eng = make_engine(params)
    try:
        q2, f2, s2 = eng.contig(hb)
        if check_columns:
            cols, c2 = want[3], eng.columns()
            assert len(cols) == len(c2), (len(cols), len(c2))
            for nm in cols.dtype.names:
                bad = np.nonzero(cols[nm] != c2[nm])[0]
                assert len(bad) == 0, (nm, len(bad), cols["pos"][bad[:5]], cols[nm][bad[:5]], c2[nm][bad[:5]])
        bad = np.nonzero(q != q2)[0]
        assert len(bad) == 0, ("quality bytes differ", len(bad), bad[:10], q[bad[:10]], q2[bad[:10]])
        assert np.array_equal(f, f2), "per-read flags differ"
        d = np.nonzero(s != s2)[0]
        assert len(d) == 0, ("stats differ", d, s[d], s2[d])
        assert_coded_download(eng, hb, q2, f2, s2)
    finally:
        eng.close()
    return q, f, s


def assert_coded_download(eng, hb, q, f, s):
    """The coded download (crumble_hip_encode / _download_coded + the host-side expansion) of the run that `eng` holds is
    byte for byte the plain download (q, f, s): whole batch on several threads, and through the cursor from arbitrary
    starts (as the BAM writer walks it read by read, skipping some)."""
    import ctypes as C
    c, f3, s3 = eng.download_coded()
    assert c.n_bases == hb.n_bases and c.n_words == (hb.n_bases + 15) // 16 and c.n_blocks == (hb.n_bases + 4095) // 4096
    assert np.array_equal(f, f3) and np.array_equal(s, s3), "coded download: flags / statistics differ"
    got = eng.decode(c, hb.quals, threads=3)
    bad = np.nonzero(got != q)[0]
    assert len(bad) == 0, ("coded download differs from the plain one", len(bad), bad[:10], got[bad[:10]], q[bad[:10]], list(c.value), c.n_escapes)
    if hb.n_reads:
        # read by read from a random start, in place (input bytes overwritten), every third read skipped
        rng = np.random.default_rng(hb.n_bases)
        r0 = int(rng.integers(0, hb.n_reads))
        r1 = min(hb.n_reads, r0 + 300)
        buf = hb.quals.copy()
        cur = ca.DecodeCursor()
        eng.L.crumble_decode_seek(C.byref(c), int(hb.seq_off[r0]), C.byref(cur))
        for r in range(r0, r1):
            a, b = int(hb.seq_off[r]), int(hb.seq_off[r + 1])
            skip = (r % 3) == 1
            eng.L.crumble_decode_run(C.byref(c), C.byref(cur), b - a, buf.ctypes.data + a, None if skip else buf.ctypes.data + a)
            want = hb.quals[a:b] if skip else q[a:b]
            assert np.array_equal(buf[a:b], want), ("cursor decode differs", r, skip)
        assert cur.base == int(hb.seq_off[r1])


def cigar_zoo(cases, reps=60, seed=7):
    """Overlapping reads over a zoo of CIGAR shapes, random mapq/flags/quals, keep-BED."""
    from crumble_amd import HostBatch
    rng = np.random.default_rng(seed)
    reads = []
    pos = 1000
    for rep in range(reps):
        for cg in cases:
            ops = ca.parse_cigar(cg)
            lq = int(sum(int(o >> 4) for o in ops if (int(o) & 15) in (0, 1, 4, 7, 8)))
            seq = "".join("ACGTN"[i] for i in rng.choice(5, lq, p=[.3, .2, .2, .28, .02]))
            reads.append(dict(pos=pos, cigar=cg, seq=seq, qual=rng.integers(2, 42, lq).astype(np.uint8),
                              mapq=int(rng.choice([0, 3, 20, 60, 255])),
                              flag=int(rng.choice([0, 16, 1024, 4], p=[.6, .3, .05, .05]))))
            pos += int(rng.integers(0, 3))
    # zero-length SEQ read and a read ending in a deletion (fenced paths)
    reads.append(dict(pos=pos, cigar="10M", seq="", qual=np.zeros(0, np.uint8)))
    reads.append(dict(pos=pos, cigar="6M2D", seq="ACGTAC", qual=30))
    return HostBatch.from_reads(reads, bed=[(1010, 1030), (1100, 1101)])


def str_stress_batch(n_reads=400, seed=9):
    """Reads made of tandem repeats, homopolymers, near-repeats and random stretches (lengths 1..700):
    every branch of find_STR / add_rep (str_finder.c:34-189) gets exercised through the mask_LC seam."""
    rng = np.random.default_rng(seed)
    reads, pos = [], 100
    for i in range(n_reads):
        L = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 64, 150, 251, 499, 501, 520, 700]))
        seq = []
        while len(seq) < L:
            kind = rng.integers(0, 5)
            if kind == 0:
                seq += list(rng.choice(list("ACGT"), int(rng.integers(1, 40))))
            elif kind == 1:
                seq += [str(rng.choice(list("ACGTN")))] * int(rng.integers(2, 60))
            else:
                unit = list(rng.choice(list("ACGT"), int(rng.integers(1, 10))))
                rep = unit * int(rng.integers(2, 20))
                if kind == 3 and len(rep) > 4:                  # imperfect copy
                    rep[int(rng.integers(0, len(rep)))] = str(rng.choice(list("ACGT")))
                if kind == 4:                                   # partial last copy
                    rep = rep[:len(rep) - int(rng.integers(0, len(unit)))]
                seq += rep
        seq = "".join(seq[:L])
        reads.append(dict(pos=pos, cigar=f"{L}M", seq=seq, qual=rng.integers(2, 60, L), mapq=60))
        pos += int(rng.integers(0, 30))
    return HostBatch.from_reads(reads)


def pblock_stress(engine_factory, path, seed=3, n_reads=400):
    """P-block seam (snp:803-834) on raw quality strings of every flavour: constant stretches with outliers,
    binned values, drifting random walks (no jump but a range > 2*level), uniform noise; read lengths
    0..300 so that segments start and end anywhere in the 16-byte chunks.  Returns the number of reads."""
    import oracle_lib as O
    rng = np.random.default_rng(seed)
    if True:
        for level, qcap in ((8, 40), (1, 93), (2, 20), (16, 40), (40, 60), (70, 93), (3, 35)):
            quals, so = [], [0]
            for r in range(n_reads):
                L = int(rng.choice([0, 1, 2, 15, 16, 17, 31, 33, 64, 100, 150, 151, 300]))
                kind = rng.integers(0, 5)
                if kind == 0:
                    q = np.full(L, int(rng.integers(2, 94)))
                    if L:
                        k = rng.integers(0, L, max(1, L // 20)); q[k] = rng.integers(2, 94, len(k))
                elif kind == 1:
                    q = rng.choice([2, 11, 25, 37], L, p=[.05, .1, .25, .6])
                elif kind == 2:
                    q = np.clip(30 + np.cumsum(rng.integers(-level, level + 1, L)), 0, 93)
                elif kind == 3:
                    q = rng.integers(0, 94, L)
                else:
                    q = np.clip(40 + np.cumsum(rng.integers(-2, 3, L)), 0, 93)
                quals.append(np.asarray(q, np.uint8)); so.append(so[-1] + L)
            q = np.concatenate(quals) if quals else np.zeros(0, np.uint8)
            so = np.asarray(so, np.int64)
            eng = engine_factory(ca.preset_params(9, pblock=level, qcap=qcap))
            eng.set_option("pblock", path)
            try:
                got = eng.pblock(so, q)
            finally:
                eng.close()
            for r in range(n_reads):
                a, b = so[r], so[r + 1]
                want = O.pblock(q[a:b], level, qcap)
                assert np.array_equal(got[a:b], want), (path, level, qcap, r, q[a:b].tolist(), got[a:b].tolist(), want.tolist())
    return n_reads


KCOL_VARIANTS = ("v1", "v1perm", "v2")     # crumble_hip_set_option("kcol", ...): every column kernel must give the same bits


def gap_batch(tid=0):
    """Five clusters of reads with coverage gaps between them and ONE read (10M5980D10M) that spans every gap: the
    columns under its deletion belong to no cluster's owned reads (round-2 advisor finding: windows dropped them from
    the statistics and from the carried depth average)."""
    rng = np.random.default_rng(17)
    seq = "".join("ACGT"[i] for i in rng.integers(0, 4, 6100))
    reads = [dict(pos=1000, cigar="10M5980D10M", seq=seq[:10] + seq[5990:6000], qual=30, mapq=60)]
    for c in range(5):
        start = 1100 + c * 1200
        for i in range(40):
            p = start + int(rng.integers(0, 300))
            reads.append(dict(pos=p, cigar="100M", seq=seq[p - 1000:p - 900], qual=int(rng.integers(20, 40)), mapq=60))
    reads[1:] = sorted(reads[1:], key=lambda r: r["pos"])
    return HostBatch.from_reads(reads, tid=tid)





def pblock_long_stress(engine_factory, path, seed=1, n_cases=40):
    """P-block (snp:803-834) on long-read-shaped quality strings: reads of up to 30 kb, one dominant value with sparse
    preserved / outlying values (HiFi after the replay), slow drifts, noise; -y pbccs, no preserved value, and MORE than
    four preserved values (the table path); levels 1..70.  Reads longer than the word-parallel kernels' 4 KiB tile,
    runs that cross hundreds of 64-byte words, data that ends anywhere in a tile."""
    import oracle_lib as O
    rng = np.random.default_rng(seed)
    for it in range(n_cases):
        tech = int(rng.integers(3))
        level = int(rng.choice([1, 2, 4, 8, 16, 30, 70]))
        p = ca.preset_params(9, tech="pbccs", pblock=level) if tech == 0 else ca.preset_params(9, pblock=level)
        if tech == 2:
            for v in rng.choice([5, 20, 21, 22, 40, 59, 60, 75, 93], int(rng.integers(1, 8)), replace=False):
                p.preserve_qual[int(v)] = 1
        nr = int(rng.integers(1, 12))
        lens = rng.integers(1, 9000, nr) if rng.random() < 0.7 else rng.integers(1, 200, nr)
        if rng.random() < 0.3:
            lens[rng.integers(nr)] = int(rng.integers(4000, 30000))
        so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        n = int(so[-1])
        mode = int(rng.integers(4))
        if mode == 0:
            q = rng.choice([60, 93, 5, 56, 40, 59, 70, 90], n, p=[.85, .07, .01, .02, .01, .02, .01, .01])
        elif mode == 1:
            q = np.clip(60 + np.cumsum(rng.integers(-3, 4, n)), 0, 93)
        elif mode == 2:
            q = rng.integers(0, 94, n)
        else:
            q = np.full(n, 60)
            k = rng.integers(0, n, max(1, n // 50))
            q[k] = rng.choice([93, 20, 45, 75], len(k))
        q = q.astype(np.uint8)
        pq = np.frombuffer(bytes(p.preserve_qual), np.uint8)
        want = np.concatenate([O.pblock(q[so[i]:so[i + 1]], p.pblock, p.qcap, pq) for i in range(nr)])
        eng = engine_factory(p)
        eng.set_option("pblock", path)
        try:
            got = eng.pblock(so, q)
        finally:
            eng.close()
        d = np.flatnonzero(got != want)
        assert len(d) == 0, (it, path, tech, level, mode, lens.tolist(), d[:5].tolist(), got[d[:5]].tolist(), want[d[:5]].tolist())

_GPU_COUNT = None


def gpu_count():
    """Visible GPUs, asked of torch in a CHILD process: torch brings its own copy of the HIP runtime, and a process
    that has libcrumble_hip.so's (/opt/rocm) copy loaded as well dies in the runtimes' exit handlers ("double free or
    corruption", exit code 134, after every test has passed)."""
    global _GPU_COUNT
    if _GPU_COUNT is None:
        import subprocess, sys
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        try:
            _GPU_COUNT = int(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            _GPU_COUNT = 0
    return _GPU_COUNT
