"""crumble_stream (include/crumble_hip.h; csrc/stream_kernels.h, stream_host.h) and the device pipeline of crumble-hip
(csrc/cli.c, CRUMBLE_HIP_PIPELINE=device): a BAM file's records stay on the device from inflate to deflate.

What is checked, against the host walk of the same file and against the oracle:
  * the record starts the device finds (speculative chunk walk) are the starts a sequential walk finds, for feeds cut at any
    BGZF member (so records straddle feeds), and the contig runs are the file's;
  * the engine fed from the device-resident records (k_st_fields / k_st_gather) writes the oracle's qualities into the
    records, every other byte of the stream is the input's;
  * what drain() hands back are valid gzip members (zlib checks CRC-32 and ISIZE) of exactly the records;
  * crumble-hip with the device pipeline writes the uncompressed stream the host pipeline writes, BED keep-regions included;
  * a truncated record, bytes that are not BAM records and out-of-order contigs are refused; a CIGAR behind a CG tag makes
    the driver take the host pipeline.
The emulated tier runs the literal kernel sources on the CPU; the GPU tier runs the shipped binary and library."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import bamio, gz
import oracle_lib as O
import synth
from test_bamio import ROOT, BGZF_EOF, three_groups, py_bam, py_bgzf, walk_bam, _oracle_file, _file_quals


def host_walk(raw):
    """(header bytes, record offsets) of an inflated BAM stream, sequentially"""
    l_text = struct.unpack("<i", raw[4:8])[0]
    p = 8 + l_text
    n_ref = struct.unpack("<i", raw[p:p + 4])[0]
    p += 4
    for _ in range(n_ref):
        ln = struct.unpack("<i", raw[p:p + 4])[0]
        p += 4 + ln + 4
    hdr, offs = p, []
    while p < len(raw):
        offs.append(p)
        p += 4 + struct.unpack("<i", raw[p:p + 4])[0]
    return n_ref, hdr, offs


def record_quals(stream):
    """the quality bytes of a headerless record stream, and the stream with them blanked"""
    q, p, blank = [], 0, bytearray(stream)
    while p < len(stream):
        bsz, = struct.unpack("<i", stream[p:p + 4])
        r = p + 4
        lname = stream[r + 8]
        nco, = struct.unpack("<H", stream[r + 12:r + 14])
        ls, = struct.unpack("<i", stream[r + 16:r + 20])
        qo = r + 32 + lname + 4 * nco + (ls + 1) // 2
        q.append(np.frombuffer(stream[qo:qo + ls], np.uint8))
        blank[qo:qo + ls] = bytes(ls)
        p += 4 + bsz
    return (np.concatenate(q) if q else np.zeros(0, np.uint8)), bytes(blank)


HALVES = []


def run_stream(library, engine, path, cuts, level=6, bed=None, small_out=False):
    """the file through a Stream in len(cuts) + 1 feeds (cuts: member indices); returns the inflated output records"""
    comp = np.fromfile(path, np.uint8)
    raw = gzip.decompress(comp.tobytes())
    n_ref, hdr, offs = host_walk(raw)
    blk, consumed, total = gz.bgzf_scan(comp, library)
    assert consumed == len(comp) and total == len(raw)
    edges = [0] + [int(blk["coff"][c]) - 18 for c in cuts] + [len(comp)]
    s = gz.Stream(n_ref, hdr, 0, library)
    if cuts:
        s.reserve(len(comp) // 2, len(raw) // 3)         # a hint that is too small: the buffers grow
    out, seen_groups, done_bytes = b"", [], hdr
    try:
        for i in range(len(edges) - 1):
            last = i == len(edges) - 2
            s.feed(comp[edges[i]:edges[i + 1]], last)
            buf, ro = s.peek()
            # the buffer holds the stream from the first undrained byte on (the header is dropped); behind what is processed and
            # waits (pending), the device's record starts are the sequential walk's
            base = done_bytes
            assert s.pending() == 0 and buf.tobytes() == raw[base:base + len(buf)]
            want = [o - base for o in offs if base <= o and o - base + 4 <= len(buf) and
                    o - base + 4 + struct.unpack("<i", raw[o:o + 4])[0] <= len(buf)]
            assert list(ro) == want
            gs = s.groups()
            k = 0
            for j, g in enumerate(gs):
                if not g.complete:
                    assert j == len(gs) - 1 and not last
                    break
                sel = None
                if bed is not None and g.tid >= 0:
                    sel = bed[bed[:, 0] == g.tid][:, 1:]
                s.process(engine, j, sel)
                seen_groups.append((g.tid, g.n_reads))
                k += 1
            if k:
                nbytes = gs[k - 1].byte_end
                assert gs[0].byte_start == 0
                if small_out:                        # an output buffer of two members: the rest waits at the front of the buffer
                    cap, kk, back = 2 * (0xff00 + 31) + 5, k, b""
                    m = s.drain(kk, level, cap, flush=False).tobytes() if nbytes < 0xff00 else b"x"
                    assert nbytes >= 0xff00 or (m == b"" and s.pending() == nbytes)        # less than a member and no flush: it waits
                    kk = 0 if nbytes < 0xff00 else kk
                    while True:
                        m = s.drain(kk, level, cap).tobytes()
                        kk = 0
                        back += gzip.decompress(m) if m else b""
                        assert len(m) <= cap and s.pending() == nbytes - len(back)
                        if not s.pending():
                            break
                        assert len(gzip.decompress(m)) == 2 * 0xff00
                    assert [(g.tid, g.byte_start) for g in s.groups()[:1]] in ([], [(gs[k].tid, 0)] if k < len(gs) else [])
                elif k > 1:                          # the drain in two halves (crumble-hip processes the next runs in between)
                    HALVES.append(k)
                    taken, cap = s.drain_begin(k - 1, level)
                    assert taken == gs[k - 2].byte_end
                    with pytest.raises(ca.CrumbleError):
                        s.drain(0, level)                # (no other drain, no feed, while one is in flight)
                    m = s.drain_end(cap).tobytes()
                    back = gzip.decompress(m) if m else b""
                    m = s.drain(1, level).tobytes()      # (the runs are numbered anew: the last one is run 0 now)
                    back += gzip.decompress(m) if m else b""
                else:
                    m = s.drain(k, level).tobytes()
                    back = gzip.decompress(m) if m else b""
                assert len(back) == nbytes
                out += back
                done_bytes += nbytes
        assert not s.groups() and done_bytes == len(raw)
    finally:
        s.close()
    return raw[hdr:], out, seen_groups


def check_against_oracle(params, path, records_in, records_out, bed=None):
    got, blank_out = record_quals(records_out)
    _, blank_in = record_quals(records_in)
    assert blank_out == blank_in                      # names, positions, CIGARs, bases, tags: untouched
    want = _oracle_file(params, path, bed)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]


def _cases(library, make_engine, tmp_path, scale):
    refs = [("chrI", 60000), ("chrII", 25000)]
    bs = [synth.generate(synth.config("ce", scale=scale)), synth.generate(synth.config("ce", scale=scale / 2, seed=78, tid=1)), three_groups()[1][2]]
    path = str(tmp_path / "in.bam")
    bamio.write_batches(path, refs, bs, level=6)
    nblk = len(gz.bgzf_scan(np.fromfile(path, np.uint8), library)[0])
    p5 = ca.preset_params(5)
    eng = make_engine(p5)
    try:
        for cuts in ([nblk // 2], sorted({1, nblk // 3, 2 * nblk // 3} - {0})) if nblk > 3 else ([],):     # (one feed: the keep-region case below)
            rin, rout, groups = run_stream(library, eng, path, cuts, small_out=len(cuts) == 1)
            assert groups == [(0, bs[0].n_reads), (1, bs[1].n_reads), (-1, bs[2].n_reads)]
            check_against_oracle(p5, path, rin, rout)
        assert HALVES                               # (the two-halves drain was exercised)
    finally:
        eng.close()
    # keep regions
    (tmp_path / "keep.bed").write_text("chrI\t1000\t3000\nchrII\t500\t800\n")
    rd = bamio.Reader(path); bed = rd.load_bed(tmp_path / "keep.bed"); rd.close()
    p1 = ca.preset_params(1)
    eng = make_engine(p1)
    try:
        rin, rout, _ = run_stream(library, eng, path, [], bed=bed)
        check_against_oracle(p1, path, rin, rout, bed)
    finally:
        eng.close()


def _refusals(library, make_engine, tmp_path):
    refs = [("chrI", 60000), ("chrII", 25000)]
    rec = lambda tid, pos, i: dict(tid=tid, pos=pos, cigar=[(40 << 4)], seq=[1, 2, 4, 8] * 10, qual=[30] * 40, name=f"r{i}")
    # a record cut short at the end of the file
    raw = py_bam(refs, [rec(0, 100 + i, i) for i in range(20)])
    n_ref, hdr, offs = host_walk(raw)
    for bad, what in ((raw[:-7], "truncated"), (raw[:hdr] + bytes(range(256)) * 40, "BAM record"),
                      (raw[:offs[5]] + struct.pack("<i", 20) + raw[offs[5] + 4:], "BAM record"),
                      (py_bam(refs, [rec(1, 100, 0), rec(0, 100, 1)]), "order")):
        comp = np.frombuffer(py_bgzf(bad)[:-28], np.uint8)
        s = gz.Stream(n_ref, hdr, 0, library)
        with pytest.raises(ca.CrumbleError) as ei:
            s.feed(comp, True)
        assert what in str(ei.value), (what, str(ei.value))
        s.close()
    # a contig run that may go on is not processed
    comp = np.frombuffer(py_bgzf(raw)[:-28], np.uint8)
    s = gz.Stream(n_ref, hdr, 0, library)
    s.feed(comp, False)
    eng = make_engine(ca.preset_params(9))
    gs = s.groups()
    assert len(gs) == 1 and not gs[0].complete and gs[0].n_reads == 20
    with pytest.raises(ca.CrumbleError):
        s.process(eng, 0)
    s.feed(np.zeros(0, np.uint8), True)
    assert s.groups()[0].complete
    s.process(eng, 0)
    assert gzip.decompress(s.drain(1, 1).tobytes())[:-0 or None] != b""
    eng.close(); s.close()


def _wrong_guesses(library, make_engine, tmp_path):
    """records built so that FOUR BYTES IN FRONT of every record start there is a plausible header too (the last quality bytes
    as a block_size below 2^29, the record's own block_size as a refID below the reference count, its pos -- 1 mod 256 -- as a
    one-byte name ...): every chunk's speculative guess is wrong, the chain check repairs it, the result is the sequential
    walk's (found with 300 and 1500 real contigs: a refID test is weak once there are hundreds of references)"""
    refs = [(f"c{t}", 3000000) for t in range(200)]
    recs = [dict(tid=7, pos=1 + 256 * i, cigar=[(50 << 4)], seq=[(1, 2, 4, 8)[(i + k) % 4] for k in range(50)], qual=[20] * 50, name=f"q{i % 10}xyz")
            for i in range(3000)]
    raw = py_bam(refs, recs)
    n_ref, hdr, offs = host_walk(raw)
    assert len(raw) - hdr > 4 * 65536
    comp = np.frombuffer(py_bgzf(raw)[:-28], np.uint8)
    s = gz.Stream(n_ref, hdr, 0, library)
    p9 = ca.preset_params(9)
    eng = make_engine(p9)
    try:
        s.feed(comp, True)
        buf, ro = s.peek()
        assert list(ro) == [o - hdr for o in offs]
        gs = s.groups()
        assert [(g.tid, g.n_reads) for g in gs] == [(7, 3000)]
        s.process(eng, 0)
        out = gzip.decompress(s.drain(1, 1).tobytes())
    finally:
        eng.close(); s.close()
    (tmp_path / "fp.bam").write_bytes(py_bgzf(raw))
    check_against_oracle(p9, tmp_path / "fp.bam", raw[hdr:], out)


def _cli(exe, tmp_path, scale, env=None, timeout=1800, chunk=66000, level=6, quick=False, reblock=0):
    refs = [("chrI", 60000), ("chrII", 25000)]
    bs = [synth.generate(synth.config("ce", scale=scale)), synth.generate(synth.config("ce", scale=scale / 2, seed=78, tid=1)), three_groups()[1][2]]
    bamio.write_batches(tmp_path / "in.bam", refs, bs, level=level)
    if reblock:                                              # small members: a small read buffer then cuts the file many times
        (tmp_path / "in.bam").write_bytes(py_bgzf(gzip.decompress(open(tmp_path / "in.bam", "rb").read()), block=reblock))
    (tmp_path / "keep.bed").write_text("chrI\t1000\t3000\nchrII\t500\t800\n")
    rd = bamio.Reader(tmp_path / "in.bam"); bed = rd.load_bed(tmp_path / "keep.bed"); rd.close()
    for args, params, b in ((["-5"], ca.preset_params(5), None), (["-1", "-R", str(tmp_path / "keep.bed")], ca.preset_params(1), bed)):
        st, err = {}, {}
        for mode in ("host", "device"):
            out = tmp_path / f"out_{mode}.bam"
            r = subprocess.run([exe, "-v", "-z", *args, str(tmp_path / "in.bam"), str(out)], env=dict(os.environ, CRUMBLE_HIP_PIPELINE=mode, **(env or {})),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
            assert r.returncode == 0, r.stderr
            assert ("pipeline: device" in r.stderr) == (mode == "device") and "Gbases/s file to file" in r.stderr
            data = open(out, "rb").read()
            assert data[-28:] == BGZF_EOF
            st[mode] = gzip.decompress(data)
            err[mode] = [l for l in r.stderr.splitlines() if l[:1].islower() and l.split()[-1].isdigit() and len(l.split()) == 2]
        assert st["device"] == st["host"]
        assert err["device"] == err["host"] and len(err["host"]) >= 20          # the statistics block
        # the same with the file cut into many feeds and drains (members carried from one read buffer to the next, contig runs
        # that end in a later feed, both buffers of each kind in use)
        if b is not None and quick:
            continue
        r = subprocess.run([exe, "-z", *args, str(tmp_path / "in.bam"), str(tmp_path / "out_cut.bam")], timeout=timeout, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, CRUMBLE_HIP_PIPELINE="device", CRUMBLE_HIP_DP_CHUNK=str(chunk), CRUMBLE_HIP_DP_PIECE="20000", CRUMBLE_HIP_ASYNC_DRAIN="1", **(env or {})))
        assert r.returncode == 0, r.stderr
        assert gzip.decompress(open(tmp_path / "out_cut.bam", "rb").read()) == st["host"]
        assert np.array_equal(_file_quals(tmp_path / "out_device.bam"), _oracle_file(params, tmp_path / "in.bam", b))
    # records without qualities (0xff, SAMv1 4.2.3) between ordinary ones, the @PG line of the default run
    raw = py_bam(refs, [dict(tid=0, pos=100 + 2 * i, cigar=[(40 << 4)], seq=[(1, 2, 4, 8)[(k * 5 + k // 7) % 4] for k in range(40)],
                             qual=(None if i % 4 == 1 else [20 + (i * 7 + k) % 21 for k in range(40)]), name=f"m{i}") for i in range(300)])
    (tmp_path / "noq.bam").write_bytes(py_bgzf(raw))
    outs = {}
    for mode in ("host", "device"):
        r = subprocess.run([exe, "-9", str(tmp_path / "noq.bam"), str(tmp_path / "noq_out.bam")], env=dict(os.environ, CRUMBLE_HIP_PIPELINE=mode, **(env or {})),
                           stderr=subprocess.PIPE, text=True, timeout=timeout)
        assert r.returncode == 0, r.stderr
        text, _, recs = walk_bam(tmp_path / "noq_out.bam")
        assert "@PG\tID:crumble-hip" in text
        outs[mode] = [bytes(x["qual"]) for x in recs]
    assert outs["host"] == outs["device"] and all(q == b"\xff" * 40 for i, q in enumerate(outs["device"]) if i % 4 == 1)
    assert any(q != bytes(20 + (i * 7 + k) % 21 for k in range(40)) for i, q in enumerate(outs["device"]) if i % 4 != 1)
    # a CIGAR behind a CG tag (SAMv1 4.2.2: placeholder kSmN + CG:B,I): the device pipeline hands the file to the host pipeline
    real = [(2 << 4) | 0, (1 << 4) | 1, (3 << 4) | 2, (1 << 4) | 0]
    cgtag = b"NMC\x01CGBI" + struct.pack("<I4I", 4, *real) + b"XSZz\0"
    recs = [dict(tid=0, pos=1 + i, cigar=[(40 << 4)], seq=[1, 2, 4, 8] * 10, qual=[30 + i % 5] * 40, name=f"r{i}") for i in range(30)]
    recs.insert(3, dict(tid=0, pos=4, cigar=[(4 << 4) | 4, (6 << 4) | 3], seq=[1, 2, 4, 8], qual=[1, 2, 3, 4], aux=cgtag))
    (tmp_path / "cg.bam").write_bytes(py_bgzf(py_bam(refs, recs)))
    st = {}
    for mode in ("host", "device"):
        r = subprocess.run([exe, "-v", "-z", "-9", str(tmp_path / "cg.bam"), str(tmp_path / f"cg_{mode}.bam")], env=dict(os.environ, CRUMBLE_HIP_PIPELINE=mode, **(env or {})),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        assert r.returncode == 0, r.stderr
        assert ("the host pipeline runs" in r.stderr) == (mode == "device")
        st[mode] = gzip.decompress(open(tmp_path / f"cg_{mode}.bam", "rb").read())
    assert st["host"] == st["device"] and len(walk_bam(tmp_path / "cg_device.bam")[2]) == 31


# ------------------------------------------------------------------------------------------ emulated tier (no GPU)
def test_emu_stream_records_stay_on_the_device(tmp_path):
    import emu_lib
    _cases(emu_lib.load(), emu_lib.engine, tmp_path, 0.25)


def test_emu_stream_refusals(tmp_path):
    import emu_lib
    _refusals(emu_lib.load(), emu_lib.engine, tmp_path)


def test_emu_stream_wrong_guesses_repaired(tmp_path):
    import emu_lib
    _wrong_guesses(emu_lib.load(), emu_lib.engine, tmp_path)


def test_cli_device_pipeline_emulated(tmp_path):
    import emu_lib
    emu_lib.load()
    _cli(os.path.join(ROOT, "tests", "emu", "crumble-hip-emu"), tmp_path, 0.15, env={"CRUMBLE_HIP_CODEC": "host"}, quick=True, reblock=3000, chunk=5000)


def test_cli_pipeline_option_checked(tmp_path):
    import emu_lib
    emu_lib.load()
    exe = os.path.join(ROOT, "tests", "emu", "crumble-hip-emu")
    bamio.write_batches(tmp_path / "none.bam", [("chrI", 60000)], [])
    r = subprocess.run([exe, "-5", str(tmp_path / "none.bam"), str(tmp_path / "o.bam")], env=dict(os.environ, CRUMBLE_HIP_PIPELINE="gpu"),
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "CRUMBLE_HIP_PIPELINE" in r.stderr


# ------------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
def test_gpu_stream_records_stay_on_the_device(product, tmp_path):
    _cases(product, lambda p: ca.Engine(p, 0), tmp_path, 1.0)


@pytest.mark.gpu
def test_gpu_stream_wrong_guesses_repaired(product, tmp_path):
    _wrong_guesses(product, lambda p: ca.Engine(p, 0), tmp_path)


@pytest.mark.gpu
def test_gpu_stream_refusals(product, tmp_path):
    _refusals(product, lambda p: ca.Engine(p, 0), tmp_path)


@pytest.mark.gpu
def test_cli_device_pipeline_gpu(tmp_path):
    _cli(os.path.join(ROOT, "crumble_amd", "crumble-hip"), tmp_path, 1.0, timeout=600)


@pytest.mark.gpu
def test_cli_device_pipeline_many_contigs_gpu(tmp_path):
    """forty contigs of very different sizes and an unmapped tail, several megabytes of records: the device pipeline's file
    is the host pipeline's (level 1 and the default level) and the oracle's"""
    exe = os.path.join(ROOT, "crumble_amd", "crumble-hip")
    rng = np.random.default_rng(5)
    refs, bs = [], []
    for tid in range(40):
        ln = int(rng.choice([5000, 20000, 150000, 400000]))
        refs.append((f"c{tid}", ln))
        if tid % 7 == 3:
            continue                                            # a contig without reads
        cfg = synth.config("ce", scale=1.0, contig_len=max(4500, ln // 20), seed=100 + tid, tid=tid, depth=float(rng.choice([3.0, 12.0, 40.0])))
        bs.append(synth.generate(cfg))
    bs.append(three_groups()[1][2])
    bamio.write_batches(tmp_path / "in.bam", refs, bs, level=1)
    st = {}
    for mode, level in (("host", "1"), ("device", "1"), ("device", "-1"), ("device", "6")):
        out = tmp_path / f"out_{mode}{level}.bam"
        cut = {"CRUMBLE_HIP_DP_CHUNK": "200000", "CRUMBLE_HIP_DP_PIECE": "300000", "CRUMBLE_HIP_ASYNC_DRAIN": "1"} if level == "6" else {}      # many feeds, many drains (left in flight)
        r = subprocess.run([exe, "-v", "-z", "-9", str(tmp_path / "in.bam"), str(out)], env=dict(os.environ, CRUMBLE_HIP_PIPELINE=mode, CRUMBLE_BAM_LEVEL=level, **cut),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        st[(mode, level)] = gzip.decompress(open(out, "rb").read())
    assert st[("host", "1")] == st[("device", "1")] == st[("device", "-1")] == st[("device", "6")]
    assert np.array_equal(_file_quals(tmp_path / "out_device1.bam"), _oracle_file(ca.preset_params(9), tmp_path / "in.bam"))


@pytest.mark.gpu
def test_cli_device_pipeline_a_thousand_contigs_gpu(tmp_path):
    """more contig runs in one feed than crumble_stream_groups hands out at a time (256), most of them a handful of reads
    (hs38DH has 3171 decoys and alts of a few kb), contigs without reads between them"""
    exe = os.path.join(ROOT, "crumble_amd", "crumble-hip")
    rng = np.random.default_rng(9)
    refs = [(f"d{t}", 9000) for t in range(1000)]
    recs = []
    for t in range(1000):
        if t % 13 == 5:
            continue
        for i in range(int(rng.integers(1, 12))):
            L = int(rng.integers(30, 120))
            recs.append(dict(tid=t, pos=100 + 40 * i + int(rng.integers(0, 30)), cigar=[(L << 4)], seq=rng.choice([1, 2, 4, 8], L).tolist(),
                             qual=rng.integers(2, 42, L).tolist(), name=f"r{t}_{i}", mapq=int(rng.choice([0, 20, 60]))))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    (tmp_path / "in.bam").write_bytes(py_bgzf(py_bam(refs, recs)))
    st = {}
    for mode in ("host", "device"):
        r = subprocess.run([exe, "-v", "-z", "-9", str(tmp_path / "in.bam"), str(tmp_path / f"out_{mode}.bam")], env=dict(os.environ, CRUMBLE_HIP_PIPELINE=mode),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert ("pipeline: device" in r.stderr) == (mode == "device")
        st[mode] = gzip.decompress(open(tmp_path / f"out_{mode}.bam", "rb").read())
    assert st["host"] == st["device"] and len(walk_bam(tmp_path / "out_device.bam")[2]) == len(recs)
    assert np.array_equal(_file_quals(tmp_path / "out_device.bam"), _oracle_file(ca.preset_params(9), tmp_path / "in.bam"))
