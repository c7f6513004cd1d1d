"""ctypes mirror of include/crumble_hip.h (the C ABI) and batch helpers.

Python is only the test / benchmark driver: the product is
``crumble_amd/csrc`` (HIP kernels + C host layer) behind the C ABI.
"""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

ST_N = 288
ST_HIST0 = 32
ST_NAMES = [
    "columns", "het_A", "het_qual_A", "hom_A", "hom_qual_A", "discrep_A",
    "het_B", "het_qual_B", "hom_B", "hom_qual_B", "discrep_B", "diff", "indel",
    "indel_qual", "low_mqual_perc", "clip_perc", "ins_len_perc",
    "indel_ov_perc", "over_depth", "reads", "bases", "bases_changed",
    "triggers", "too_deep",
]
RF_KEEP, RF_NOPBLOCK, RF_PASSTHRU = 1, 2, 4


ABI = 6          # CRUMBLE_HIP_ABI of include/crumble_hip.h: the struct layouts below are that version's


class Params(C.Structure):
    """crumble_params (snp_score.c:185-226 hot-path fields)."""
    _fields_ = [
        ("reduce_qual", C.c_int32), ("binary_qual", C.c_int32),
        ("iSTR_add", C.c_int32), ("sSTR_add", C.c_int32),
        ("iSTR_mul", C.c_double), ("sSTR_mul", C.c_double),
        ("qlow", C.c_int32), ("qcutoff", C.c_int32), ("qhigh", C.c_int32),
        ("qcap", C.c_int32), ("min_mqual", C.c_int32),
        ("min_qual_A", C.c_int32), ("min_indel_A", C.c_int32),
        ("min_qual_B", C.c_int32), ("min_indel_B", C.c_int32),
        ("pblock", C.c_int32), ("softclip", C.c_int32), ("perfect_col", C.c_int32),
        ("indel_fract", C.c_double), ("min_discrep_A", C.c_double),
        ("min_discrep_B", C.c_double), ("low_mqual_perc", C.c_double),
        ("clip_perc", C.c_double), ("ins_len_perc", C.c_double),
        ("over_depth", C.c_double), ("indel_ov_perc", C.c_double),
        ("preserve_qual", C.c_uint8 * 256),
    ]


class Batch(C.Structure):
    """crumble_batch: one contig of reads in SoA form."""
    _fields_ = [
        ("tid", C.c_int32), ("n_reads", C.c_int64),
        ("pos", C.c_void_p), ("mapq", C.c_void_p), ("flag", C.c_void_p),
        ("seq_off", C.c_void_p), ("cig_off", C.c_void_p), ("cigar", C.c_void_p),
        ("bases", C.c_void_p), ("quals", C.c_void_p),
        ("n_bed", C.c_int32), ("bed_start", C.c_void_p), ("bed_end", C.c_void_p),
        ("bases4", C.c_void_p),
        ("l_qseq", C.c_void_p), ("n_cigar", C.c_void_p), ("n_bases", C.c_int64), ("n_cigar_ops", C.c_int64),
        ("qualsp", C.c_void_p), ("qual_bits", C.c_int32), ("qual_lut", C.c_uint8 * 16),
    ]


class Coded(C.Structure):
    """crumble_coded: the result as 2 bits per base + escape bytes (include/crumble_hip.h, "coded download")."""
    _fields_ = [
        ("n_bases", C.c_int64), ("n_words", C.c_int64), ("n_blocks", C.c_int64), ("n_escapes", C.c_int64),
        ("value", C.c_uint8 * 2), ("pad_", C.c_uint8 * 6), ("cap", C.c_uint8 * 256),
        ("codes", C.c_void_p), ("esc_off", C.c_void_p), ("esc", C.c_void_p),
    ]


class DecodeCursor(C.Structure):
    _fields_ = [("base", C.c_int64), ("esc_pos", C.c_int64)]


class Column(C.Structure):
    """crumble_column."""
    _fields_ = [
        ("pos", C.c_int32), ("n_plp", C.c_int32), ("score", C.c_int32),
        ("call1", C.c_uint8), ("call2", C.c_uint8), ("preserve", C.c_uint8),
        ("perfect", C.c_uint8), ("keep_qual", C.c_uint8), ("trigger", C.c_uint8),
        ("window", C.c_uint8), ("skipped", C.c_uint8), ("min_pos2", C.c_int32),
    ]


COLUMN_DTYPE = np.dtype([
    ("pos", "<i4"), ("n_plp", "<i4"), ("score", "<i4"),
    ("call1", "u1"), ("call2", "u1"), ("preserve", "u1"), ("perfect", "u1"),
    ("keep_qual", "u1"), ("trigger", "u1"), ("window", "u1"), ("skipped", "u1"),
    ("min_pos2", "<i4")])
assert COLUMN_DTYPE.itemsize == C.sizeof(Column) == 24

NT16 = "=ACMGRSVTWYHKDBN"
NT16_CODE = {c: i for i, c in enumerate(NT16)}
CIG_OPS = "MIDNSHP=X"
_CIG_RE = re.compile(r"(\d+)([MIDNSHP=X])")


def parse_cigar(s):
    """'5S10M2I' -> uint32 BAM ops (len<<4|op)."""
    if s in ("*", ""):
        return np.zeros(0, np.uint32)
    return np.array([(int(n) << 4) | CIG_OPS.index(o) for n, o in _CIG_RE.findall(s)],
                    np.uint32)


class HostBatch:
    """Owns the numpy arrays behind a crumble_batch."""

    def __init__(self, tid, pos, mapq, flag, seq_off, cig_off, cigar, bases, quals,
                 bed=None):
        self.tid = int(tid)
        self.pos = np.ascontiguousarray(pos, np.int32)
        self.mapq = np.ascontiguousarray(mapq, np.uint8)
        self.flag = np.ascontiguousarray(flag, np.uint16)
        self.seq_off = np.ascontiguousarray(seq_off, np.int64)
        self.cig_off = np.ascontiguousarray(cig_off, np.int64)
        self.cigar = np.ascontiguousarray(cigar, np.uint32)
        self.bases = np.ascontiguousarray(bases, np.uint8)
        self.quals = np.ascontiguousarray(quals, np.uint8)
        n = len(self.pos)
        assert len(self.seq_off) == n + 1 and len(self.cig_off) == n + 1
        if bed is None or len(bed) == 0:
            self.bed_start = np.zeros(0, np.int32)
            self.bed_end = np.zeros(0, np.int32)
        else:
            bed = np.asarray(bed, np.int32).reshape(-1, 2)
            self.bed_start = np.ascontiguousarray(bed[:, 0])
            self.bed_end = np.ascontiguousarray(bed[:, 1])
        # keep non-empty buffers so .ctypes.data is a valid pointer
        self._pad = np.zeros(16, np.uint8)

    @property
    def n_reads(self):
        return len(self.pos)

    @property
    def n_bases(self):
        return int(self.seq_off[-1]) if len(self.seq_off) else 0

    def _ptr(self, a):
        return a.ctypes.data if a.size else self._pad.ctypes.data

    def struct(self, packed=True):
        """crumble_batch over these arrays; packed=False always hands over the one-code-per-byte bases (the oracle and the
        BAM writer read them), packed=True the BAM-packed ones when pack4() attached them."""
        b = Batch()
        b.tid = self.tid
        b.n_reads = self.n_reads
        for name in ("pos", "mapq", "flag", "seq_off", "cig_off", "cigar", "bases",
                     "quals", "bed_start", "bed_end"):
            setattr(b, name, self._ptr(getattr(self, name)))
        b.n_bed = len(self.bed_start)
        b.bases4 = None
        if packed and getattr(self, "bases4", None) is not None:       # BAM-packed bases instead (crumble_batch.bases4)
            b.bases4 = self.bases4.ctypes.data
            b.bases = None
        if packed and getattr(self, "l_qseq", None) is not None:       # BAM's per-read lengths instead of the offset tables
            b.l_qseq, b.n_cigar = self._ptr(self.l_qseq), self._ptr(self.n_cigar)
            b.n_bases, b.n_cigar_ops = self.n_bases, int(self.cig_off[-1]) if len(self.cig_off) else 0
            b.seq_off = b.cig_off = None
        b.qualsp, b.qual_bits = None, 0
        if packed and getattr(self, "qualsp", None) is not None:       # coded qualities instead (crumble_batch.qualsp)
            b.qualsp, b.qual_bits = self.qualsp.ctypes.data, self.qual_bits
            for i in range(16):
                b.qual_lut[i] = int(self.qual_lut[i])
            b.quals = None
        return b

    def lens(self):
        """Attach BAM's own per-read lengths (crumble_batch.l_qseq / n_cigar): struct() then hands the engine those (6 bytes
        per read over PCIe) and no offset tables.  Batches with a read of more than 65535 CIGAR ops keep the offsets."""
        nc = np.diff(self.cig_off)
        if len(nc) and nc.max() > 65535:
            return self
        self.l_qseq = np.ascontiguousarray(np.diff(self.seq_off), np.int32)
        self.n_cigar = np.ascontiguousarray(nc, np.uint16)
        return self

    def pack4(self, lib):
        """Attach the bases in BAM's packing (crumble_pack_bases4): struct() then hands the engine bases4 and no bases."""
        out = np.zeros((self.n_bases >> 1) + self.n_reads + 1, np.uint8)
        lib.crumble_pack_bases4.restype = None
        lib.crumble_pack_bases4.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.crumble_pack_bases4(self.n_reads, self.seq_off.ctypes.data, self._ptr(self.bases), out.ctypes.data)
        self.bases4 = out
        return self

    def packq(self, lib):
        """Attach the qualities as 2- or 4-bit codes (crumble_pack_quals): struct() then hands the engine qualsp and no quals.
        A batch with more than sixteen distinct qualities keeps the bytes."""
        out = np.zeros((self.n_bases * 4 + 7) // 8 + 16, np.uint8)
        lut = np.zeros(16, np.uint8)
        lib.crumble_pack_quals.restype = C.c_int
        lib.crumble_pack_quals.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        bits = lib.crumble_pack_quals(self.n_bases, self._ptr(self.quals), out.ctypes.data, lut.ctypes.data)
        if bits:
            self.qualsp, self.qual_bits, self.qual_lut = out[:(self.n_bases * bits + 7) // 8 + 16], bits, lut
        return self

    @classmethod
    def from_reads(cls, reads, tid=0, bed=None):
        """reads: iterable of dicts(pos, cigar, seq, qual, mapq=60, flag=0)."""
        pos, mapq, flag, so, co = [], [], [], [0], [0]
        cig, bases, quals = [], [], []
        for r in reads:
            pos.append(r["pos"])
            mapq.append(r.get("mapq", 60))
            flag.append(r.get("flag", 0))
            ops = parse_cigar(r.get("cigar", "*"))
            cig.append(ops)
            co.append(co[-1] + len(ops))
            seq = r.get("seq", "")
            bases.append(np.array([NT16_CODE[ch] for ch in seq.upper()], np.uint8))
            q = r.get("qual", 30)
            if np.isscalar(q):
                q = np.full(len(seq), q, np.uint8)
            q = np.asarray(q, np.uint8)
            assert len(q) == len(seq)
            quals.append(q)
            so.append(so[-1] + len(seq))
        cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt))
        return cls(tid, pos, mapq, flag, so, co, cat(cig, np.uint32),
                   cat(bases, np.uint8), cat(quals, np.uint8), bed)


def stats_dict(st):
    d = {n: int(st[i]) for i, n in enumerate(ST_NAMES)}
    d["hist"] = np.asarray(st[ST_HIST0:ST_HIST0 + 256]).copy()
    return d
