"""Host ingest / egress: BGZF + BAM reader and writer, and the crumble-hip
file -> file driver.  The checker for the reader is an independent pure-Python BGZF/BAM
encoder in this file (struct + zlib from the SAMv1 text); the checker for the writer is
Python's own gzip decoder plus a pure-Python record walker."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import bamio
import synth
from crumble_amd.capi import ROOT, HostBatch

import oracle_lib as O

BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


# ---------------------------------------------------------------- independent encoder / walker
def py_bgzf(data, block=30000, empty_every=0):
    out = bytearray()
    k = 0
    for off in range(0, len(data), block):
        chunk = data[off:off + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        out += struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25)
        out += comp + struct.pack("<II", zlib.crc32(chunk), len(chunk))
        k += 1
        if empty_every and k % empty_every == 0:
            out += BGZF_EOF                                   # empty blocks may appear anywhere
    return bytes(out) + BGZF_EOF


def py_bam(refs, reads, text="@HD\tVN:1.6\tSO:coordinate\n"):
    """reads: dicts(tid,pos,mapq,flag,cigar(list of u32),seq(nt16 codes),qual(bytes or None),name,aux)."""
    out = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for n, l in refs:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    for r in reads:
        name = r.get("name", "q").encode() + b"\0"
        seq = [int(x) for x in r["seq"]]
        packed = bytes(((seq[i] << 4) | (seq[i + 1] if i + 1 < len(seq) else 0)) for i in range(0, len(seq), 2))
        qual = bytes(np.asarray(r["qual"], np.uint8)) if r.get("qual") is not None else b"\xff" * len(seq)
        aux = r.get("aux", b"")
        body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r.get("mapq", 60), 4680, len(r["cigar"]),
                           r.get("flag", 0), len(seq), -1, -1, 0)
        body += name + b"".join(struct.pack("<I", c) for c in r["cigar"]) + packed + qual + aux
        out += struct.pack("<i", len(body)) + body
    return bytes(out)


def walk_bam(path):
    """-> (header text, refs, [record dicts]) using only gzip + struct."""
    d = gzip.decompress(open(path, "rb").read())              # checks every member's CRC32 and ISIZE
    assert d[:4] == b"BAM\1"
    lt, = struct.unpack_from("<i", d, 4)
    text = d[8:8 + lt].decode()
    p = 8 + lt
    nref, = struct.unpack_from("<i", d, p); p += 4
    refs = []
    for _ in range(nref):
        ln, = struct.unpack_from("<i", d, p); p += 4
        nm = d[p:p + ln - 1].decode(); p += ln
        l, = struct.unpack_from("<i", d, p); p += 4
        refs.append((nm, l))
    recs = []
    while p < len(d):
        bs, = struct.unpack_from("<i", d, p); p += 4
        tid, pos, lname, mapq, _bin, ncig, flag, lseq, _nt, _np, _tl = struct.unpack_from("<iiBBHHHiiii", d, p)
        q = p + 32
        name = d[q:q + lname - 1]; q += lname
        cig = list(struct.unpack_from(f"<{ncig}I", d, q)); q += 4 * ncig
        sq = d[q:q + (lseq + 1) // 2]; q += (lseq + 1) // 2
        qual = d[q:q + lseq]; q += lseq
        recs.append(dict(tid=tid, pos=pos, mapq=mapq, flag=flag, cigar=cig, seqpacked=sq, qual=qual, name=name,
                         aux=d[q:p + bs]))
        p += bs
    return text, refs, recs


def batches_equal(a, b):
    for nm in ("pos", "mapq", "flag", "seq_off", "cig_off", "cigar", "bases", "quals"):
        assert np.array_equal(getattr(a, nm), getattr(b, nm)), nm
    assert a.tid == b.tid


def three_groups():
    b0 = synth.generate(synth.config("ce", scale=0.5))
    b1 = synth.generate(synth.config("ce", scale=0.2, seed=77, tid=1))
    rng = np.random.default_rng(3)
    n, L = 30, 100
    un = HostBatch(-1, np.full(n, -1), np.zeros(n), np.full(n, 4), np.arange(n + 1) * L, np.zeros(n + 1),
                   np.zeros(0, np.uint32), rng.choice([1, 2, 4, 8, 15], n * L), rng.integers(2, 42, n * L))
    return [("chrI", 60000), ("chrII", 25000)], [b0, b1, un]


# ------------------------------------------------------------------------------- tests
@pytest.fixture(autouse=True)
def _host_codec_under_emulation(request, monkeypatch):
    """crumble-hip inflates and deflates BGZF members on the device by default; emulated, a full block costs 1024 fibers some
    fifteen seconds, so the emulated driver tests keep the host codec (test_cli_device_codec_emulated below runs the device
    one on a small file; the GPU tests take the default)"""
    if "emulated" in request.node.name:
        monkeypatch.setenv("CRUMBLE_HIP_PIPELINE", "host")    # (the device pipeline, crumble-hip's default: tests/test_stream.py)
        if "device_codec" not in request.node.name:
            monkeypatch.setenv("CRUMBLE_HIP_CODEC", "host")


def test_write_batches_then_read_back(tmp_path):
    refs, bs = three_groups()
    path = tmp_path / "t.bam"
    bamio.write_batches(path, refs, bs, level=6)
    text, wrefs, recs = walk_bam(path)                        # valid gzip members, valid BAM
    assert wrefs == refs and "SO:coordinate" in text
    assert len(recs) == sum(b.n_reads for b in bs)
    assert open(path, "rb").read()[-28:] == BGZF_EOF
    assert [r["pos"] for r in recs[:bs[0].n_reads]] == list(bs[0].pos)
    rd = bamio.Reader(path, threads=3)
    assert rd.refs == refs
    got = list(rd)
    rd.close()
    assert len(got) == 3
    for a, b in zip(got, bs):
        batches_equal(a, b)


def test_both_codecs_cross_read(tmp_path):
    """BGZF blocks go through libdeflate when the host has it (dlopen) and through zlib otherwise (CRUMBLE_BAM_ZLIB=1
    forces it): a file written by either is valid gzip (Python's decoder checks every member's CRC32 and ISIZE) and is
    read back identically by the other.  The codec is a process-wide choice, so each side is its own process."""
    import subprocess
    import sys
    code = '''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ctypes as C
from crumble_amd import bamio
import synth
L = bamio.lib(); L.crumble_bam_codec.restype = C.c_char_p
mode, path, want = sys.argv[1], sys.argv[2], sys.argv[3]
assert L.crumble_bam_codec().decode() == want, L.crumble_bam_codec()
b0 = synth.generate(synth.config("ce", scale=0.5))
if mode == "write":
    bamio.write_batches(path, [("chrI", 60000)], [b0], level=int(sys.argv[4]))
else:
    got = list(bamio.Reader(path, threads=3))
    assert len(got) == 1
    for nm in ("pos", "mapq", "flag", "seq_off", "cig_off", "cigar", "bases", "quals"):
        assert np.array_equal(getattr(got[0], nm), getattr(b0, nm)), nm
print("ok")
''' % (ROOT, os.path.join(ROOT, "tests"))
    def run(env_zlib, *args):
        env = dict(os.environ)
        env.pop("CRUMBLE_BAM_ZLIB", None)
        if env_zlib:
            env["CRUMBLE_BAM_ZLIB"] = "1"
        r = subprocess.run([sys.executable, "-c", code, *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-800:]
    have_ld = bamio.lib().crumble_bam_codec is not None
    import ctypes as C
    bamio.lib().crumble_bam_codec.restype = C.c_char_p
    here = bamio.lib().crumble_bam_codec().decode()
    run(True, "write", str(tmp_path / "z.bam"), "zlib", "1")
    walk_bam(tmp_path / "z.bam")
    run(True, "read", str(tmp_path / "z.bam"), "zlib")
    if here == "libdeflate" or "CRUMBLE_BAM_ZLIB" in os.environ:
        for lvl in ("0", "1", "6"):
            run(False, "write", str(tmp_path / f"d{lvl}.bam"), "libdeflate", lvl)
            walk_bam(tmp_path / f"d{lvl}.bam")                    # an independent inflate + CRC check of every block
            run(True, "read", str(tmp_path / f"d{lvl}.bam"), "zlib")
            run(False, "read", str(tmp_path / f"d{lvl}.bam"), "libdeflate")
        run(False, "read", str(tmp_path / "z.bam"), "libdeflate")


def test_reader_against_python_encoder(tmp_path):
    rng = np.random.default_rng(11)
    refs = [("a", 5000), ("b", 9000), ("c", 100)]
    reads = []
    for tid, n in ((0, 400), (2, 7)):                         # contig b has no reads; group order = header order
        pos = np.sort(rng.integers(0, 3000 if tid == 0 else 50, n))
        for i in range(n):
            L = int(rng.integers(0, 180))                     # includes l_seq == 0 and odd lengths
            cig = [(L << 4) | 0] if L else []
            if L > 20 and rng.random() < 0.3:
                cig = [(5 << 4) | 4, ((L - 12) << 4) | 0, (3 << 4) | 2, (7 << 4) | 0]
            reads.append(dict(tid=tid, pos=int(pos[i]), mapq=int(rng.integers(0, 61)), flag=int(rng.choice([0, 16, 99, 147])),
                              cigar=cig, seq=rng.choice([1, 2, 4, 8, 15], L), qual=None if i % 50 == 7 else rng.integers(0, 94, L),
                              name=f"read{i}", aux=b"NMC\x03XSZhello\0" if i % 3 == 0 else b""))
    reads.append(dict(tid=-1, pos=-1, flag=4, cigar=[], seq=[1, 2, 4], qual=[30, 31, 32], name="un"))
    raw = py_bam(refs, reads)
    path = tmp_path / "p.bam"
    path.write_bytes(py_bgzf(raw, block=777, empty_every=5))  # records and header fields straddle blocks
    rd = bamio.Reader(path, threads=2)
    assert rd.refs == refs
    groups = list(rd)
    assert [g.tid for g in groups] == [0, 2, -1]
    k = 0
    for g in groups:
        for i in range(g.n_reads):
            r = reads[k]; k += 1
            assert (g.pos[i], g.mapq[i], g.flag[i]) == (r["pos"], r.get("mapq", 60), r.get("flag", 0))
            assert list(g.cigar[g.cig_off[i]:g.cig_off[i + 1]]) == r["cigar"]
            assert list(g.bases[g.seq_off[i]:g.seq_off[i + 1]]) == list(r["seq"])
            want = np.zeros(len(r["seq"]), np.uint8) if r.get("qual") is None else np.asarray(r["qual"], np.uint8)
            assert np.array_equal(g.quals[g.seq_off[i]:g.seq_off[i + 1]], want)
    assert k == len(reads)
    rd.close()

    # writer: names, aux, cigars and absent qualities survive; other qualities are replaced
    rd = bamio.Reader(path)
    wr = bamio.Writer(tmp_path / "o.bam", rd, pg_line="@PG\tID:test\tPN:test", level=1, threads=2)
    while True:
        g = rd.next_contig()
        if g is None:
            break
        wr.write_group((g.quals ^ 1) if g.tid != 2 else None)
    wr.close(); rd.close()
    text, wrefs, recs = walk_bam(tmp_path / "o.bam")
    assert wrefs == refs and text.endswith("@PG\tID:test\tPN:test\n")
    assert len(recs) == len(reads)
    for r, o in zip(reads, recs):
        assert o["name"] == r["name"].encode() and o["aux"] == r.get("aux", b"") and o["cigar"] == r["cigar"]
        if r.get("qual") is None:
            assert o["qual"] == b"\xff" * len(r["seq"])
        elif r["tid"] == 2:
            assert o["qual"] == bytes(np.asarray(r["qual"], np.uint8))
        else:
            assert o["qual"] == bytes(np.asarray(r["qual"], np.uint8) ^ 1)


def test_reader_rejects_bad_input(tmp_path):
    raw = py_bam([("a", 100)], [dict(tid=0, pos=1, cigar=[(3 << 4)], seq=[1, 2, 4], qual=[1, 2, 3])])
    good = py_bgzf(raw)
    cases = {
        "plain gzip": gzip.compress(raw),
        "not bam": py_bgzf(b"BAX\1" + raw[4:]),
        "truncated block": good[:len(good) - 40],
        "truncated record": py_bgzf(raw[:-2]),
        "crc": good[:30] + bytes([good[30] ^ 0x55]) + good[31:],
        "empty": b"",
    }
    for name, data in cases.items():
        p = tmp_path / "bad.bam"
        p.write_bytes(data)
        with pytest.raises(bamio.BamError):
            rd = bamio.Reader(p)
            list(rd)
    with pytest.raises(bamio.BamError):
        bamio.Reader(tmp_path / "missing.bam")
    # a CIGAR stored in the CG tag (SAMv1 4.2.2: placeholder kSmN + CG:B,I) is read from the tag
    real = [(2 << 4) | 0, (1 << 4) | 1, (3 << 4) | 2, (1 << 4) | 0]          # 2M1I3D1M: 4 query bases, 6 reference bases
    cgtag = b"NMC\x01CGBI" + struct.pack("<I4I", 4, *real) + b"XSZz\0"
    raw = py_bam([("a", 100)], [dict(tid=0, pos=1, cigar=[(4 << 4) | 4, (6 << 4) | 3], seq=[1, 2, 4, 8], qual=[1, 2, 3, 4], aux=cgtag),
                                dict(tid=0, pos=5, cigar=[(3 << 4) | 4, (50 << 4) | 3], seq=[1, 2, 4], qual=[1, 2, 3])])   # no CG tag: a real 3S50N
    (tmp_path / "cg.bam").write_bytes(py_bgzf(raw))
    g, = list(bamio.Reader(tmp_path / "cg.bam"))
    assert list(g.cig_off) == [0, 4, 6] and list(g.cigar) == real + [(3 << 4) | 4, (50 << 4) | 3]


def test_load_bed_matches_bed_collapse(tmp_path):
    refs = [("chrI", 60000), ("chrII", 25000)]
    bamio.write_batches(tmp_path / "h.bam", refs, [])
    (tmp_path / "k.bed").write_text("# c\ntrack x\nbrowser y\n\nchrII\t10\t20\nchrI 500 900\nchrI\t100\t600\textra\nchrI\t2000\t2100\n")
    rd = bamio.Reader(tmp_path / "h.bam")
    got = rd.load_bed(tmp_path / "k.bed")
    assert got.tolist() == [[0, 100, 900], [0, 2000, 2100], [1, 10, 20]]
    (tmp_path / "bad.bed").write_text("chrIII\t1\t2\n")
    with pytest.raises(bamio.BamError):
        rd.load_bed(tmp_path / "bad.bed")
    rd.close()


# ------------------------------------------------------------------------ purge_tags (f3)
def py_aux_split(aux):
    """[(tag, whole field bytes)] with the SAMv1 4.2.4 type sizes."""
    out, p = [], 0
    fixed = {b"A": 1, b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4, b"d": 8}
    while p < len(aux):
        t = aux[p + 2:p + 3]
        if t in fixed:
            e = p + 3 + fixed[t]
        elif t in (b"Z", b"H"):
            e = aux.index(b"\0", p + 3) + 1
        else:
            assert t == b"B"
            n, = struct.unpack_from("<I", aux, p + 4)
            e = p + 8 + n * fixed[aux[p + 3:p + 4]]
        out.append((aux[p:p + 2], aux[p:e]))
        p = e
    return out


def py_purge(aux, white=None, black=None, bd=None, bi=None):
    """purge_tags, snp_score.c:989-1054, restated."""
    keep = []
    for tag, field in py_aux_split(aux):
        if white is not None:
            if tag not in white:
                continue
        elif black is not None and tag in black:
            continue
        for name, b in ((b"BD", bd), (b"BI", bi)):
            if b and tag == name and field[2:3] == b"Z":
                field = field[:3] + bytes((b[2] if c >= b[1] else b[0]) for c in field[3:-1]) + b"\0"
        keep.append(field)
    return b"".join(keep)


def test_purge_tags(tmp_path):
    rng = np.random.default_rng(5)
    refs = [("a", 100000)]
    fields = [b"NMC\x05", b"XSZhello world\0", b"BDZ" + bytes(rng.integers(33, 90, 40).astype(np.uint8)) + b"\0",
              b"BIZ" + bytes(rng.integers(33, 90, 40).astype(np.uint8)) + b"\0", b"ASi" + struct.pack("<i", -7),
              b"XFf" + struct.pack("<f", 1.5), b"ZBBS" + struct.pack("<I3H", 3, 1, 2, 3), b"MDZ10A5\0", b"RGZgrp\0",
              b"XHH1AE3\0", b"XAA!", b"xds" + struct.pack("<h", -2), b"BDZ\0"]
    reads = []
    for i in range(300):
        L = int(rng.integers(1, 80))
        k = rng.permutation(len(fields))[:int(rng.integers(0, len(fields)))]
        reads.append(dict(tid=0, pos=10 * i, cigar=[(L << 4)], seq=rng.choice([1, 2, 4, 8], L), qual=rng.integers(2, 41, L),
                          name=f"r{i}", aux=b"".join(fields[j] for j in k)))
    (tmp_path / "a.bam").write_bytes(py_bgzf(py_bam(refs, reads), block=5000))
    cases = [dict(whitelist="NM,RG,BD"), dict(blacklist="XS,BD,ZB"), dict(whitelist="NM", blacklist="NM"),
             dict(bd=(40 + 33, 20 + 33, 50 + 33)), dict(blacklist="XS,MD", bd=(35, 60, 80), bi=(36, 50, 70)), dict()]
    for kw in cases:
        rd = bamio.Reader(tmp_path / "a.bam")
        wr = bamio.Writer(tmp_path / "o.bam", rd, level=1, threads=3)
        wr.set_aux_filter(**kw)
        while rd.next_contig() is not None:
            wr.write_group(None)
        wr.close(); rd.close()
        _, _, recs = walk_bam(tmp_path / "o.bam")
        assert len(recs) == len(reads)
        def split(l):
            if l is None:
                return None
            l = l.replace(",", "")
            return {l[i:i + 2].encode() for i in range(0, len(l), 2)}
        for r, o in zip(reads, recs):
            want = py_purge(r["aux"], split(kw.get("whitelist")), split(kw.get("blacklist")), kw.get("bd"), kw.get("bi"))
            assert o["aux"] == want, (kw, r["aux"], o["aux"], want)
            assert o["name"] == r["name"].encode() and o["qual"] == bytes(np.asarray(r["qual"], np.uint8))
    rd = bamio.Reader(tmp_path / "a.bam")
    wr = bamio.Writer(tmp_path / "o.bam", rd)
    for bad in (dict(whitelist="NMX"), dict(bd=(40, 0, 0)), dict(bi=(0, 50, 0))):
        with pytest.raises(bamio.BamError):
            wr.set_aux_filter(**bad)
    wr.close(); rd.close()


def _oracle_file(params, path, bed=None):
    """What crumble-hip must write: per group, the oracle's bytes (the checker)."""
    O.load()
    rd = bamio.Reader(path)
    out = []
    for g in rd:
        if bed is not None and g.tid >= 0:
            sel = bed[bed[:, 0] == g.tid][:, 1:]
            g = HostBatch(g.tid, g.pos, g.mapq, g.flag, g.seq_off, g.cig_off, g.cigar, g.bases, g.quals, sel)
        out.append(O.contig(params, g)[0])
    rd.close()
    return np.concatenate(out)


def _file_quals(path):
    return np.frombuffer(b"".join(r["qual"] for r in walk_bam(path)[2]), np.uint8)


def test_cli_end_to_end_emulated(tmp_path):
    """crumble-hip (csrc/cli.c) linked against the CPU-emulated kernels: file in, file out, bytes == oracle."""
    import emu_lib
    emu_lib.load()
    exe = os.path.join(ROOT, "tests", "emu", "crumble-hip-emu")
    refs, bs = three_groups()
    # the second contig's qualities as a binned instrument writes them (eight values; the first keeps its 22: CRUMBLE_HIP_CODED_QUALS
    # below then sends one contig as codes and one as bytes)
    bs[1].quals = np.array([2, 11, 17, 22, 27, 33, 37, 40], np.uint8)[bs[1].quals % 8]
    bamio.write_batches(tmp_path / "in.bam", refs, bs)
    (tmp_path / "keep.bed").write_text("chrI\t1000\t3000\nchrII\t500\t800\n")
    for args, params, bed in (
            (["-5"], ca.preset_params(5), None),
            (["-1", "-u", "35", "-K", "2-3", "-R", str(tmp_path / "keep.bed")], ca.preset_params(1, really_keep="2-3", qhigh=35), True),
            (["-9", "-y", "pbccs", "-S"], ca.preset_params(9, tech="pbccs", softclip=1), None)):
        r = subprocess.run([exe, "-v", *args, str(tmp_path / "in.bam"), str(tmp_path / "out.bam")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert "Gbases/s file to file" in r.stderr
        bedarr = None
        if bed:
            rd = bamio.Reader(tmp_path / "in.bam"); bedarr = rd.load_bed(tmp_path / "keep.bed"); rd.close()
        want = _oracle_file(params, tmp_path / "in.bam", bedarr)
        got = _file_quals(tmp_path / "out.bam")
        assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
        text, wrefs, recs = walk_bam(tmp_path / "out.bam")
        assert wrefs == refs and "@PG\tID:crumble-hip" in text and len(recs) == sum(b.n_reads for b in bs)
        # the default run took the coded download (2 bits per base + escapes, expanded into the records by the writer);
        # one byte per base back from the device gives the same file
        r = subprocess.run([exe, *args, str(tmp_path / "in.bam"), str(tmp_path / "out_plain.bam")], env=dict(os.environ, CRUMBLE_HIP_PLAIN_DOWNLOAD="1"),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert np.array_equal(_file_quals(tmp_path / "out_plain.bam"), want)
        # ... and so do the qualities sent to the device as 2- / 4-bit codes where a contig has at most sixteen values
        # (CRUMBLE_HIP_CODED_QUALS=1: crumble_pack_quals + crumble_batch.qualsp; contigs with more values send bytes)
        r = subprocess.run([exe, *args, str(tmp_path / "in.bam"), str(tmp_path / "out_cq.bam")], env=dict(os.environ, CRUMBLE_HIP_CODED_QUALS="1"),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert np.array_equal(_file_quals(tmp_path / "out_cq.bam"), want)
    # records without qualities (0xff, SAMv1 4.2.3) between ordinary ones: the coded expansion skips exactly their bases
    raw = py_bam(refs, [dict(tid=0, pos=100 + 2 * i, cigar=[(40 << 4)], seq=[(1, 2, 4, 8)[(k * 5 + k // 7) % 4] for k in range(40)],
                             qual=(None if i % 4 == 1 else [20 + (i * 7 + k) % 21 for k in range(40)]), name=f"m{i}") for i in range(300)])
    (tmp_path / "noq.bam").write_bytes(py_bgzf(raw))
    outs = []
    for env in ({}, {"CRUMBLE_HIP_PLAIN_DOWNLOAD": "1"}):
        r = subprocess.run([exe, "-9", str(tmp_path / "noq.bam"), str(tmp_path / "noq_out.bam")], env=dict(os.environ, **env), stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        outs.append([bytes(x["qual"]) for x in walk_bam(tmp_path / "noq_out.bam")[2]])
    assert outs[0] == outs[1] and all(q == b"\xff" * 40 for i, q in enumerate(outs[0]) if i % 4 == 1)
    assert any(q != bytes(20 + (i * 7 + k) % 21 for k in range(40)) for i, q in enumerate(outs[0]) if i % 4 != 1)
    # purge_tags through the command line (-T drops, -e/-f/-g bins BD) leaves names and qualities alone
    raw = py_bam(refs, [dict(tid=0, pos=100 + i, cigar=[(50 << 4)], seq=[1] * 50, qual=[30] * 50, name=f"t{i}",
                             aux=b"NMC\x01BDZ" + bytes([40 + i, 70]) + b"\0XSZab\0") for i in range(5)])
    (tmp_path / "tags.bam").write_bytes(py_bgzf(raw))
    r = subprocess.run([exe, "-9", "-T", "XS", "-e", "2", "-f", "20", "-g", "40", str(tmp_path / "tags.bam"), str(tmp_path / "t2.bam")],
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    recs = walk_bam(tmp_path / "t2.bam")[2]
    assert [x["aux"] for x in recs] == [b"NMC\x01BDZ" + bytes([35 if 40 + i < 53 else 73, 73]) + b"\0" for i in range(5)]
    # options that were not rebuilt are refused loudly
    r = subprocess.run([exe, "-r", "chrI", str(tmp_path / "in.bam"), str(tmp_path / "o2.bam")], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "does not provide" in r.stderr


def test_cli_device_codec_emulated(tmp_path):
    """crumble-hip with its default codec -- BGZF members inflated (crumble_gz_inflate behind the reader's hook) and deflated
    (crumble_gz_deflate behind the writer's) by the device kernels, emulated here -- writes the records the host codec writes:
    same uncompressed stream, valid gzip members; each direction alone as well"""
    import emu_lib
    emu_lib.load()
    exe = os.path.join(ROOT, "tests", "emu", "crumble-hip-emu")
    refs = [("chrI", 60000), ("chrII", 25000)]
    bs = [synth.generate(synth.config("ce", scale=0.06)), synth.generate(synth.config("ce", scale=0.03, seed=78, tid=1))]
    bamio.write_batches(tmp_path / "in.bam", refs, bs, level=6)
    streams = {}
    for mode in ("host", "gpu", "gpu-inflate", "gpu-deflate"):
        out = tmp_path / f"out_{mode}.bam"
        r = subprocess.run([exe, "-v", "-5", "-z", str(tmp_path / "in.bam"), str(out)], env=dict(os.environ, CRUMBLE_HIP_CODEC=mode),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        want_dev = {"host": (False, False), "gpu": (True, True), "gpu-inflate": (True, False), "gpu-deflate": (False, True)}[mode]
        line = [l for l in r.stderr.splitlines() if l.startswith("BGZF codec:")][0]
        assert ("inflate device" in line) == want_dev[0] and ("deflate device" in line) == want_dev[1], line
        streams[mode] = gzip.decompress(open(out, "rb").read())           # every member's CRC-32 and ISIZE checked
        assert open(out, "rb").read()[-28:] == BGZF_EOF
    assert streams["gpu"] == streams["host"] == streams["gpu-inflate"] == streams["gpu-deflate"]
    want = _oracle_file(ca.preset_params(5), tmp_path / "in.bam")
    assert np.array_equal(_file_quals(tmp_path / "out_gpu.bam"), want)
    r = subprocess.run([exe, "-5", str(tmp_path / "in.bam"), str(tmp_path / "o.bam")], env=dict(os.environ, CRUMBLE_HIP_CODEC="zlib"),
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "CRUMBLE_HIP_CODEC" in r.stderr


def test_cli_multi_device_emulated(tmp_path):
    """crumble-hip -j 2 (csrc/cli.c: contigs LPT-packed onto per-device engines, one worker thread per engine, a
    reader and a writer thread, groups detached from the reader) writes byte for byte what -j 1 writes, reports
    the same summed statistics and is bit-exact against the oracle -- here with both engines on the one emulated
    device (CRUMBLE_HIP_DEVICES=0,0: the statistics are then added on the host instead of by RCCL)."""
    import emu_lib
    emu_lib.load()
    exe = os.path.join(ROOT, "tests", "emu", "crumble-hip-emu")
    refs = [("chrI", 400000), ("chrII", 20000), ("chrIII", 90000), ("chrIV", 150000), ("chrV", 5000)]
    bs = []
    for tid, (_, ln) in enumerate(refs):
        cfg = synth.config("ce", scale=1.0, contig_len=max(4500, ln // 60), seed=77 + tid, tid=tid, depth=6.0)
        bs.append(synth.generate(cfg))
    # a read that spans coverage gaps: the columns under its deletion are counted by -w's windows too (round-2 advisor)
    from parity_util import gap_batch
    refs.append(("chrGap", 8000)); bs.append(gap_batch(tid=len(refs) - 1))
    bs.append(HostBatch.from_reads([dict(pos=0, cigar="*", seq="ACGTAC", qual=[9, 30, 31, 12, 40, 2], flag=4)], tid=-1))
    bamio.write_batches(tmp_path / "in.bam", refs, bs)
    outs = {}
    for j in ("1", "2", "3", "8"):
        env = dict(os.environ, CRUMBLE_HIP_DEVICES=",".join("0" * int(j)))
        r = subprocess.run([exe, "-v", "-v", "-5", "-j", j, "-b", str(tmp_path / f"s{j}.bed"), str(tmp_path / "in.bam"), str(tmp_path / f"out{j}.bam")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr
        outs[j] = (open(tmp_path / f"out{j}.bam", "rb").read(), open(tmp_path / f"s{j}.bed").read(),
                   [l for l in r.stderr.splitlines() if l.split() and l.split()[0] in ("columns", "bases", "bases_changed", "triggers", "reads")])
        if j != "1":
            assert f"{j} device(s)" in r.stderr
    text1 = walk_bam(tmp_path / "out1.bam")
    for j in ("2", "3", "8"):
        t = walk_bam(tmp_path / f"out{j}.bam")
        assert t[1] == text1[1] and [x["qual"] for x in t[2]] == [x["qual"] for x in text1[2]] and len(t[2]) == len(text1[2])
        assert outs[j][1] == outs["1"][1] and outs[j][2] == outs["1"][2]
    want = _oracle_file(ca.preset_params(5), tmp_path / "in.bam", None)
    assert np.array_equal(_file_quals(tmp_path / "out2.bam"), want)
    # -w: every contig cut into windows over the engines, same bytes, -b lines and statistics
    env = dict(os.environ, CRUMBLE_HIP_DEVICES="0,0,0")
    r = subprocess.run([exe, "-v", "-5", "-j", "3", "-w", "-b", str(tmp_path / "sw.bed"), str(tmp_path / "in.bam"), str(tmp_path / "outw.bam")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(_file_quals(tmp_path / "outw.bam"), want)
    assert open(tmp_path / "sw.bed").read() == outs["1"][1]
    assert [l for l in r.stderr.splitlines() if l.split() and l.split()[0] in ("columns", "bases", "bases_changed", "triggers", "reads")] == outs["1"][2]


def _oracle_bed(params, path):
    """The -b lines the reference would print (snp_score.c:1496,1676,1768,1802,1810), from the oracle's trace."""
    O.load()
    rd = bamio.Reader(path)
    names = [n for n, _ in rd.refs]
    lines = []
    for g in rd:
        if g.tid < 0:
            continue
        cols = O.contig(params, g, trace=True)[3]
        for c in cols:
            why = ["VDEEP"] if c["skipped"] == 2 else [] if c["skipped"] == 1 else \
                [nm for bit, nm in ((0x10, "DEEP"), (0x20, "CLIP"), (0x40, "INDEL_LEN"), (0x80, "INDEL_COVERAGE")) if c["skipped"] & bit]
            for w in why:
                lines.append(f"{names[g.tid]}\t{max(int(c['pos']) - 50, 0)}\t{int(c['pos']) + 50}\t{w}\n")
    rd.close()
    return lines


def test_cli_suspect_bed_emulated(tmp_path):
    """-b: the suspicious-region BED equals what the oracle's column trace implies, line for line."""
    import emu_lib
    emu_lib.load()
    exe = os.path.join(ROOT, "tests", "emu", "crumble-hip-emu")
    refs, bs = three_groups()
    bamio.write_batches(tmp_path / "in.bam", refs, bs)
    args = ["-5", "-P", "1.3", "-C", "0.02", "-Z", "0.05", "-V", "0.6"]
    params = ca.preset_params(5, over_depth=1.3, clip_perc=0.02, ins_len_perc=0.05, indel_ov_perc=0.6)
    r = subprocess.run([exe, *args, "-b", str(tmp_path / "s.bed"), str(tmp_path / "in.bam"), str(tmp_path / "out.bam")],
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = open(tmp_path / "s.bed").readlines()
    want = _oracle_bed(params, tmp_path / "in.bam")
    assert len(want) > 20 and {l.split("\t")[3] for l in want} >= {"DEEP\n", "CLIP\n"}
    assert got == want


def test_htslib_backend_absent_is_a_clean_error(tmp_path):
    """SAM / CRAM and -I / -O go through a system libhts resolved with dlopen.  There is none in
    this image (and none may be stood in for): the library says so (crumble_hts_available 0, CRUMBLE_BAM_ENOHTS),
    the driver stops with exit 1 and a message, nothing is written, and BAM keeps working (every other test)."""
    import ctypes as C
    L = bamio.lib()
    L.crumble_hts_available.argtypes = [C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(64)
    if L.crumble_hts_available(buf, 64):
        pytest.skip(f"a system htslib {buf.value.decode()} is present")
    err = C.c_int(0)
    L.crumble_hts_open.restype = C.c_void_p
    L.crumble_hts_open.argtypes = [C.c_char_p] * 7 + [C.c_int, C.POINTER(C.c_int)]
    assert L.crumble_hts_open(b"x.sam", None, b"y.sam", None, None, None, None, 1, C.byref(err)) is None
    assert err.value == -24 and b"libhts" in L.crumble_bam_last_error()
    import emu_lib
    emu_lib.load()
    exe = os.path.join(ROOT, "tests", "emu", "crumble-hip-emu")
    refs, bs = three_groups()
    bamio.write_batches(tmp_path / "in.bam", refs, bs[:1])
    (tmp_path / "in.sam").write_text("@HD\tVN:1.6\n@SQ\tSN:chrI\tLN:1000\nr1\t0\tchrI\t10\t60\t4M\t*\t0\t0\tACGT\tIIII\n")
    for args in (["-I", "bam", str(tmp_path / "in.bam"), str(tmp_path / "o.bam")],
                 ["-O", "cram,version=3.0", str(tmp_path / "in.bam"), str(tmp_path / "o.cram")],
                 [str(tmp_path / "in.bam"), str(tmp_path / "o.cram")],
                 [str(tmp_path / "in.sam"), str(tmp_path / "o.bam")]):
        r = subprocess.run([exe, "-5", *args], stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and "libhts" in r.stderr, (args, r.stderr)
        assert not os.path.exists(args[-1])


def test_htslib_backend_round_trip_when_present(tmp_path):
    """Only where a system libhts exists (none in the build image): the layout self-check of hts_load() passed
    (crumble_hts_available), and BAM -> SAM -> BAM through the htslib backend carries the same records as BAM -> BAM
    through the native reader / writer, qualities equal to the oracle's."""
    import ctypes as C
    L = bamio.lib()
    L.crumble_hts_available.argtypes = [C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(64)
    if not L.crumble_hts_available(buf, 64):
        pytest.skip("no system htslib")
    import emu_lib
    emu_lib.load()
    exe = os.path.join(ROOT, "tests", "emu", "crumble-hip-emu")
    refs, bs = three_groups()
    bamio.write_batches(tmp_path / "in.bam", refs, bs)
    for a, b_ in (("in.bam", "o1.sam"), ("o1.sam", "o2.bam")):
        r = subprocess.run([exe, "-5" if a == "in.bam" else "-z", "-z", str(tmp_path / a), str(tmp_path / b_)], stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "-5", "-z", str(tmp_path / "in.bam"), str(tmp_path / "native.bam")], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    t1, t2 = walk_bam(tmp_path / "o2.bam"), walk_bam(tmp_path / "native.bam")
    assert len(t1[2]) == len(t2[2])
    # (the second pass re-crumbles already quantised qualities; what must agree is everything but the qualities)
    for x, y in zip(t1[2], t2[2]):
        assert {k: v for k, v in x.items() if k != "qual"} == {k: v for k, v in y.items() if k != "qual"}


# ---------------------------------------------------------------- the htslib backend, executed through a test double
FAKEHTS = os.path.join(ROOT, "tests", "fakehts")


def _fakehts(kind):
    r = subprocess.run(["make", "-C", FAKEHTS], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    d = os.path.join(FAKEHTS, kind)
    return dict(os.environ, LD_LIBRARY_PATH=d + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))


def _sam_text(refs, batches):
    """SAM text of HostBatches (names r<i>; mate fields empty)."""
    out = ["@HD\tVN:1.6\tSO:coordinate"] + [f"@SQ\tSN:{n}\tLN:{l}" for n, l in refs]
    k = 0
    for hb in batches:
        for i in range(hb.n_reads):
            ops = hb.cigar[hb.cig_off[i]:hb.cig_off[i + 1]]
            cg = "".join(f"{int(o) >> 4}{'MIDNSHP=X'[int(o) & 15]}" for o in ops) or "*"
            sq = "".join("=ACMGRSVTWYHKDBN"[int(b)] for b in hb.bases[hb.seq_off[i]:hb.seq_off[i + 1]]) or "*"
            ql = "".join(chr(int(q) + 33) for q in hb.quals[hb.seq_off[i]:hb.seq_off[i + 1]]) or "*"
            rn = refs[hb.tid][0] if hb.tid >= 0 else "*"
            out.append(f"r{k}\t{int(hb.flag[i])}\t{rn}\t{int(hb.pos[i]) + 1}\t{int(hb.mapq[i])}\t{cg}\t*\t0\t0\t{sq}\t{ql}\tNM:i:{k % 300}\tXT:Z:t{k}")
            k += 1
    return "\n".join(out) + "\n"


def _sam_quals(path):
    return [l.split("\t")[10] for l in open(path) if not l.startswith("@")]


def _run_fakehts_round_trip(exe, tmp_path):
    """in.sam -> crumble-hip -5 -> out.sam through the htslib backend with the test double on LD_LIBRARY_PATH: hts_load(),
    hts_layout_check(), the bam1_t -> SoA conversion, the quality write-back, -I / -O parsing, purge_tags and the @PG
    line all execute, and the output qualities are the oracle's.  This pins NO parity with a real htslib (there is
    none here, and the double was written from the same reading of the headers as the backend): it turns "never
    executed" into "executed".  A double that lays bam1_core_t out differently must be refused."""
    import oracle_lib as O
    refs, bs = three_groups()
    (tmp_path / "in.sam").write_text(_sam_text(refs, bs))
    env = _fakehts("ok")
    r = subprocess.run([exe, "-5", "-I", "sam", "-O", "sam", "-T", "XT", str(tmp_path / "in.sam"), str(tmp_path / "out.sam")],
                       env=env, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    got = _sam_quals(tmp_path / "out.sam")
    p = ca.preset_params(5)
    want = []
    for hb in bs:
        wq, _, _ = O.contig(p, hb)
        want += ["".join(chr(int(q) + 33) for q in wq[hb.seq_off[i]:hb.seq_off[i + 1]]) or "*" for i in range(hb.n_reads)]
    assert got == want
    lines = [l for l in open(tmp_path / "out.sam")]
    assert any(l.startswith("@PG") and "crumble-hip" in l for l in lines)
    body = [l for l in lines if not l.startswith("@")]
    assert all("NM:i:" in l and "XT:Z:" not in l for l in body)                    # -T XT: the black list is applied on this path too
    assert [l.split("\t")[:10] for l in body] == [l.split("\t")[:10] for l in _sam_text(refs, bs).splitlines() if not l.startswith("@")]
    # by file name alone (snp_score.c:2569), and BAM input refused by the double (SAM text only): a clean error
    r = subprocess.run([exe, "-5", str(tmp_path / "in.sam"), str(tmp_path / "o2.sam")], env=env, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and _sam_quals(tmp_path / "o2.sam") == want, r.stderr
    r = subprocess.run([exe, "-5", "-O", "cram", str(tmp_path / "in.sam"), str(tmp_path / "o.cram")], env=env, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "format" in r.stderr
    # another ABI: refused by the layout self-check, nothing written
    r = subprocess.run([exe, "-5", str(tmp_path / "in.sam"), str(tmp_path / "o3.sam")], env=_fakehts("shifted"), stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "lays bam1_t out differently" in r.stderr, r.stderr
    assert not os.path.exists(tmp_path / "o3.sam")


def test_htslib_backend_through_a_test_double_emulated(tmp_path):
    import emu_lib
    emu_lib.load()
    _run_fakehts_round_trip(os.path.join(ROOT, "tests", "emu", "crumble-hip-emu"), tmp_path)


@pytest.mark.gpu
def test_htslib_backend_through_a_test_double_gpu(tmp_path):
    _run_fakehts_round_trip(os.path.join(ROOT, "crumble_amd", "crumble-hip"), tmp_path)



def test_cli_has_no_cpu_path(tmp_path):
    """Without a GPU the shipped binary stops with ENODEVICE (exit 2); it never computes on the host."""
    from parity_util import gpu_count
    if gpu_count():
        pytest.skip("GPU present")
    exe = os.path.join(ROOT, "crumble_amd", "crumble-hip")
    refs, bs = three_groups()
    bamio.write_batches(tmp_path / "in.bam", refs, bs[:1])
    r = subprocess.run([exe, "-5", str(tmp_path / "in.bam"), str(tmp_path / "out.bam")], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 2 and "no usable HIP device" in r.stderr
    assert not os.path.exists(tmp_path / "out.bam")


@pytest.mark.gpu
def test_cli_end_to_end_gpu(tmp_path):
    exe = os.path.join(ROOT, "crumble_amd", "crumble-hip")
    b0 = synth.generate(synth.config("chr20", scale=0.004))
    b1 = synth.generate(synth.config("panel", scale=0.01, tid=1))
    refs = [("chr20", 63025520), ("panel", 20000000)]
    bamio.write_batches(tmp_path / "in.bam", refs, [b0, b1])
    for args, params in ((["-5"], ca.preset_params(5)), (["-9", "-p", "4"], ca.preset_params(9, pblock=4))):
        r = subprocess.run([exe, "-v", *args, str(tmp_path / "in.bam"), str(tmp_path / "out.bam")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        want = _oracle_file(params, tmp_path / "in.bam")
        got = _file_quals(tmp_path / "out.bam")
        assert np.array_equal(got, want)


@pytest.mark.gpu
def test_cli_multi_device_and_options_gpu(tmp_path):
    """The shipped binary on the GPU box: (1) -j 2 / -j 3 == -j 1 byte for byte (qualities, -b lines, summed
    statistics) and == the oracle; with one visible GPU both engines sit on device 0 (CRUMBLE_HIP_DEVICES=0,0,
    statistics added on the host), with >= 2 visible GPUs the engines sit on devices 0 and 1 and the statistics go
    through ncclCommInitAll + crumble_hip_stats_allreduce.  (2) the host-plumbing options -t / -T, -e..-G, -z and
    -b through the real binary."""
    import ctypes
    exe = os.path.join(ROOT, "crumble_amd", "crumble-hip")
    refs = [("chrI", 4000000), ("chrII", 200000), ("chrIII", 900000), ("chrIV", 1500000), ("chrV", 50000)]
    bs = []
    for tid, (_, ln) in enumerate(refs):
        cfg = synth.config("chr20", scale=1.0, contig_len=max(4700, ln // 20), seed=31 + tid, tid=tid)
        bs.append(synth.generate(cfg))
    bs.append(HostBatch.from_reads([dict(pos=0, cigar="*", seq="ACGTAC", qual=[9, 30, 31, 12, 40, 2], flag=4)], tid=-1))
    bamio.write_batches(tmp_path / "in.bam", refs, bs)
    nd = ctypes.c_int(0)                         # (not through torch: importing it would pull in its own librccl)
    ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(nd))
    ndev = nd.value
    outs = {}
    for j in (1, 2, 3, 8):                       # (8: the device count the target names; more engines than contigs)
        devs = ",".join(str(k if ndev >= j else 0) for k in range(j))
        env = dict(os.environ, CRUMBLE_HIP_DEVICES=devs)
        r = subprocess.run([exe, "-v", "-5", "-j", str(j), "-b", str(tmp_path / f"s{j}.bed"), str(tmp_path / "in.bam"), str(tmp_path / f"out{j}.bam")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr
        stats = [l for l in r.stderr.splitlines() if l.split() and l.split()[0] in ("columns", "bases", "bases_changed", "triggers", "reads", "hom_B", "indel")]
        outs[j] = (_file_quals(tmp_path / f"out{j}.bam"), open(tmp_path / f"s{j}.bed").read(), stats)
        assert f"{j} device(s)" in r.stderr
        if j > 1 and ndev >= j:
            assert "RCCL all-reduce" in r.stderr
    want = _oracle_file(ca.preset_params(5), tmp_path / "in.bam", None)
    assert np.array_equal(outs[1][0], want)
    for j in (2, 3, 8):
        assert np.array_equal(outs[j][0], outs[1][0]) and outs[j][1] == outs[1][1] and outs[j][2] == outs[1][2]
    assert outs[1][1] == "".join(_oracle_bed(ca.preset_params(5), tmp_path / "in.bam"))
    # (1b) -w: every contig cut into windows over the engines
    for j in (2, 3, 8):
        devs = ",".join(str(k if ndev >= j else 0) for k in range(j))
        r = subprocess.run([exe, "-v", "-5", "-j", str(j), "-w", "-b", str(tmp_path / "sw.bed"), str(tmp_path / "in.bam"), str(tmp_path / "outw.bam")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=dict(os.environ, CRUMBLE_HIP_DEVICES=devs))
        assert r.returncode == 0, r.stderr
        stats = [l for l in r.stderr.splitlines() if l.split() and l.split()[0] in ("columns", "bases", "bases_changed", "triggers", "reads", "hom_B", "indel")]
        assert np.array_equal(_file_quals(tmp_path / "outw.bam"), want) and open(tmp_path / "sw.bed").read() == outs[1][1] and stats == outs[1][2]
    # (2) purge_tags and @PG through the real binary
    raw = py_bam(refs, [dict(tid=0, pos=100 + i, cigar=[(50 << 4)], seq=[1] * 50, qual=[30] * 50, name=f"t{i}",
                             aux=b"NMC\x01BDZ" + bytes([40 + i, 70]) + b"\0BIZ" + bytes([36 + i, 80]) + b"\0XSZab\0") for i in range(5)])
    (tmp_path / "tags.bam").write_bytes(py_bgzf(raw))
    r = subprocess.run([exe, "-9", "-z", "-T", "XS", "-e", "2", "-f", "20", "-g", "40", "-E", "1", "-F", "10", "-G", "50",
                        str(tmp_path / "tags.bam"), str(tmp_path / "t2.bam")], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    text, _, recs = walk_bam(tmp_path / "t2.bam")
    assert "@PG\tID:crumble-hip" not in text
    src = [b"NMC\x01BDZ" + bytes([40 + i, 70]) + b"\0BIZ" + bytes([36 + i, 80]) + b"\0XSZab\0" for i in range(5)]
    assert [x["aux"] for x in recs] == [py_purge(a, None, {b"XS"}, (35, 53, 73), (34, 43, 83)) for a in src]
    assert recs[0]["aux"] != src[0]
    r = subprocess.run([exe, "-9", "-t", "NM", str(tmp_path / "tags.bam"), str(tmp_path / "t3.bam")], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    text, _, recs = walk_bam(tmp_path / "t3.bam")
    assert "@PG\tID:crumble-hip" in text and [x["aux"] for x in recs] == [b"NMC\x01"] * 5
