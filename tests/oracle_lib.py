"""ctypes driver for oracle/liboracle.so -- the CPU checker (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from crumble_amd.capi import Params, Batch, HostBatch, COLUMN_DTYPE, ST_N, ROOT

ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_STR = os.path.join(ORACLE_DIR, "_ref", "libstrfinder_ref.so")
_o = None


def build():
    r = subprocess.run(["make", "-C", ORACLE_DIR], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        print(r.stdout)
        raise RuntimeError("building the oracle failed")


def load():
    global _o
    if _o is not None:
        return _o
    if not os.path.exists(LIB):
        build()
    L = C.CDLL(LIB)
    vp, i64 = C.c_void_p, C.c_int64
    L.crumble_oracle_contig.restype = C.c_int
    L.crumble_oracle_contig.argtypes = [C.POINTER(Params), C.POINTER(Batch), vp, vp, vp,
                                        vp, i64, C.POINTER(i64)]
    L.crumble_oracle_consensus.restype = C.c_int
    L.crumble_oracle_consensus.argtypes = [C.POINTER(Params), C.c_int, C.c_int, vp, vp, vp,
                                           vp, vp] + [C.POINTER(C.c_int)] * 4 + \
                                          [C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.crumble_oracle_pblock.restype = None
    L.crumble_oracle_pblock.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
    L.crumble_oracle_find_str.restype = C.c_int
    L.crumble_oracle_find_str.argtypes = [C.c_char_p, C.c_int, vp, vp, C.c_int]
    L.crumble_oracle_ref2query_pos.restype = C.c_int
    L.crumble_oracle_ref2query_pos.argtypes = [C.POINTER(Batch), i64, C.c_int]
    L.crumble_oracle_qpos2rpos.restype = C.c_int
    L.crumble_oracle_qpos2rpos.argtypes = [C.POINTER(Batch), i64, C.c_int]
    L.crumble_oracle_mask_lc.restype = C.c_int
    L.crumble_oracle_mask_lc.argtypes = [C.POINTER(Params), C.POINTER(Batch), i64, C.c_int,
                                         C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.crumble_oracle_adjust_lut.restype = None
    L.crumble_oracle_adjust_lut.argtypes = [vp]
    L.crumble_oracle_indel_only.restype = C.c_int
    L.crumble_oracle_indel_only.argtypes = [C.POINTER(Batch), C.c_int, C.c_int, vp]
    L.crumble_oracle_pileup.restype = i64
    L.crumble_oracle_pileup.argtypes = [C.POINTER(Batch), i64, vp, vp, vp, vp, vp]
    _o = L
    return L


def contig(params, hb: HostBatch, trace=False):
    """crumble_oracle_contig -> (quals, flags, stats[, columns])."""
    L = load()
    b = hb.struct(packed=False)
    q = np.empty(max(hb.n_bases, 1), np.uint8)
    f = np.zeros(max(hb.n_reads, 1), np.uint8)
    s = np.zeros(ST_N, np.int64)
    ntr = C.c_int64(0)
    cols = None
    cap = 0
    if trace:
        span = 0
        if hb.n_reads:
            span = int(hb.pos.max() - hb.pos.min()) + 1
        cap = span + int(hb.cigar.astype(np.int64).__rshift__(4).sum()) + 16
        cols = np.zeros(cap, COLUMN_DTYPE)
    rc = L.crumble_oracle_contig(C.byref(params), C.byref(b), q.ctypes.data, f.ctypes.data,
                                 s.ctypes.data, cols.ctypes.data if trace else None, cap,
                                 C.byref(ntr))
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    out = (q[:hb.n_bases], f[:hb.n_reads], s)
    if trace:
        out = out + (cols[:min(ntr.value, cap)],)
    return out


def consensus(params, flags, bases16, quals, mapqs, is_del=None, is_refskip=None):
    L = load()
    n = len(bases16)
    a = lambda x: np.ascontiguousarray(x, np.uint8)
    b16, q, m = a(bases16), a(quals), a(mapqs)
    d = a(is_del) if is_del is not None else np.zeros(n, np.uint8)
    rs = a(is_refskip) if is_refskip is not None else np.zeros(n, np.uint8)
    call, het, hph, ph, cp = (C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int())
    dis = C.c_float()
    L.crumble_oracle_consensus(C.byref(params), flags, n, b16.ctypes.data, q.ctypes.data,
                               m.ctypes.data, d.ctypes.data, rs.ctypes.data,
                               C.byref(call), C.byref(het), C.byref(hph), C.byref(ph),
                               C.byref(dis), C.byref(cp))
    return dict(call=call.value, het_call=het.value, het_phred=hph.value, phred=ph.value,
                discrep=dis.value, call_preserve=cp.value)


def pblock(quals, level, qcap, preserve_qual=None):
    L = load()
    q = np.array(quals, np.uint8)
    pq = np.zeros(256, np.uint8) if preserve_qual is None else np.asarray(preserve_qual, np.uint8)
    buf = np.concatenate([q, np.zeros(1, np.uint8)])
    L.crumble_oracle_pblock(buf.ctypes.data, len(q), level, qcap, pq.ctypes.data)
    return buf[:len(q)]


def find_str(s, maxn=1024):
    L = load()
    st = np.zeros(maxn, np.int32)
    en = np.zeros(maxn, np.int32)
    raw = s.encode() if isinstance(s, str) else bytes(s)
    n = L.crumble_oracle_find_str(raw + b"\0", len(raw), st.ctypes.data, en.ctypes.data, maxn)
    assert n <= maxn
    return list(zip(st[:n].tolist(), en[:n].tolist()))


class _RepEle(C.Structure):
    pass


_RepEle._fields_ = [("start", C.c_int), ("end", C.c_int),
                    ("prev", C.POINTER(_RepEle)), ("next", C.POINTER(_RepEle))]
_ref = None


def ref_find_str(s):
    """The REAL reference find_STR (str_finder.c compiled unmodified into oracle/_ref)."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_STR):
            return None
        _ref = C.CDLL(REF_STR)
        _ref.find_STR.restype = C.POINTER(_RepEle)
        _ref.find_STR.argtypes = [C.c_char_p, C.c_int, C.c_int]
    raw = s.encode() if isinstance(s, str) else bytes(s)
    buf = C.create_string_buffer(raw + b"\0" * 8)
    e = _ref.find_STR(buf, len(raw), 0)
    out = []
    while e:
        out.append((e.contents.start, e.contents.end))
        e = e.contents.next
    return out


def have_ref():
    return os.path.exists(REF_STR)


def ref2query_pos(hb, r, pos):
    b = hb.struct(packed=False)
    return load().crumble_oracle_ref2query_pos(C.byref(b), r, pos)


def qpos2rpos(hb, r, qpos):
    b = hb.struct(packed=False)
    return load().crumble_oracle_qpos2rpos(C.byref(b), r, qpos)


def mask_lc(params, hb, r, is_indel, rpos, min_pos=2**31 - 1, max_pos=0):
    b = hb.struct(packed=False)
    lo, hi = C.c_int(min_pos), C.c_int(max_pos)
    load().crumble_oracle_mask_lc(C.byref(params), C.byref(b), r, int(is_indel), rpos,
                                  C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def adjust_lut():
    lut = np.zeros(65536, np.uint8)
    load().crumble_oracle_adjust_lut(lut.ctypes.data)
    return lut.reshape(256, 256)


def pileup(hb, maxn=1 << 20):
    b = hb.struct(packed=False)
    cp = np.zeros(maxn, np.int32); rd = np.zeros(maxn, np.int64)
    qp = np.zeros(maxn, np.int32); ind = np.zeros(maxn, np.int32)
    bits = np.zeros(maxn, np.uint8)
    n = load().crumble_oracle_pileup(C.byref(b), maxn, cp.ctypes.data, rd.ctypes.data,
                                     qp.ctypes.data, ind.ctypes.data, bits.ctypes.data)
    assert n <= maxn
    return [dict(pos=int(cp[i]), read=int(rd[i]), qpos=int(qp[i]), indel=int(ind[i]),
                 is_del=bits[i] & 1, is_head=(bits[i] >> 1) & 1, is_tail=(bits[i] >> 2) & 1,
                 is_refskip=(bits[i] >> 3) & 1) for i in range(n)]


def indel_only(hb, dist=20, q=37):
    b = hb.struct(packed=False)
    out = np.empty(max(hb.n_bases, 1), np.uint8)
    rc = load().crumble_oracle_indel_only(C.byref(b), dist, q, out.ctypes.data)
    assert rc == 0
    return out[:hb.n_bases]
