"""ctypes view of the device BGZF codec (crumble_gz_* of include/crumble_hip.h): what the tests and bench.py call.
The product's users are crumble-hip's reader and writer (csrc/cli.c installs it into libcrumble_bamio's codec hooks)."""
import ctypes as C

import numpy as np

from .engine import CrumbleError, lib


class GzBlock(C.Structure):
    _fields_ = [("coff", C.c_int64), ("uoff", C.c_int64), ("clen", C.c_uint32), ("ulen", C.c_uint32),
                ("crc", C.c_uint32), ("pad_", C.c_uint32)]


BLOCK_DTYPE = np.dtype([("coff", "<i8"), ("uoff", "<i8"), ("clen", "<u4"), ("ulen", "<u4"), ("crc", "<u4"), ("pad_", "<u4")])


def _sig(L):
    vp, i64 = C.c_void_p, C.c_int64
    L.crumble_gz_create.restype = vp
    L.crumble_gz_create.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.crumble_gz_destroy.restype = None
    L.crumble_gz_destroy.argtypes = [vp]
    L.crumble_bgzf_scan.restype = i64
    L.crumble_bgzf_scan.argtypes = [vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)]
    L.crumble_gz_inflate.restype = C.c_int
    L.crumble_gz_inflate.argtypes = [vp, vp, i64, vp, i64, vp, i64, C.c_int]
    L.crumble_gz_deflate_bound.restype = i64
    L.crumble_gz_deflate_bound.argtypes = [i64]
    L.crumble_gz_deflate.restype = C.c_int
    L.crumble_gz_deflate.argtypes = [vp, vp, i64, C.c_int, vp, i64, C.POINTER(i64)]
    L.crumble_gz_set_option.restype = C.c_int
    L.crumble_gz_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.crumble_gz_times.restype = C.c_int
    L.crumble_gz_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return L


def bgzf_scan(data, library=None):
    """(blocks as a numpy record array, bytes of whole members walked, total ISIZE) of a BGZF byte string / uint8 array"""
    L = _sig(library if library is not None else lib())
    a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
    consumed, total = C.c_int64(0), C.c_int64(0)
    n = L.crumble_bgzf_scan(a.ctypes.data, a.size, None, 0, C.byref(consumed), C.byref(total))
    if n < 0:
        raise ValueError("not BGZF")
    blk = np.zeros(max(int(n), 1), BLOCK_DTYPE)
    n2 = L.crumble_bgzf_scan(a.ctypes.data, a.size, blk.ctypes.data, n, C.byref(consumed), C.byref(total))
    assert n2 == n
    return blk[:n], int(consumed.value), int(total.value)


class GzCodec:
    def __init__(self, device=0, library=None):
        self.L = _sig(library if library is not None else lib())
        err = C.c_int(0)
        self.h = self.L.crumble_gz_create(device, C.byref(err))
        if not self.h:
            raise CrumbleError(err.value, "crumble_gz_create", self.L)

    def set_option(self, name, value):
        rc = self.L.crumble_gz_set_option(self.h, name.encode(), value.encode())
        if rc:
            raise CrumbleError(rc, "crumble_gz_set_option", self.L)

    def close(self):
        if self.h:
            self.L.crumble_gz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def inflate(self, data, check_crc=True, blocks=None):
        """the concatenated members of a BGZF byte string inflated on the device -> uint8 array"""
        a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
        if blocks is None:
            blocks, consumed, total = bgzf_scan(a, self.L)
            if consumed != a.size:
                raise ValueError("trailing bytes behind the last whole BGZF member")
        else:
            total = int(blocks["uoff"][-1] + blocks["ulen"][-1]) if len(blocks) else 0
        out = np.empty(max(total, 1), np.uint8)
        blocks = np.ascontiguousarray(blocks)
        rc = self.L.crumble_gz_inflate(self.h, a.ctypes.data, a.size, blocks.ctypes.data, len(blocks), out.ctypes.data, total,
                                       1 if check_crc else 0)
        if rc:
            raise CrumbleError(rc, "crumble_gz_inflate", self.L)
        return out[:total]

    def deflate(self, data, level=6):
        a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
        cap = int(self.L.crumble_gz_deflate_bound(a.size))
        out = np.empty(max(cap, 1), np.uint8)
        n = C.c_int64(0)
        rc = self.L.crumble_gz_deflate(self.h, a.ctypes.data, a.size, level, out.ctypes.data, cap, C.byref(n))
        if rc:
            raise CrumbleError(rc, "crumble_gz_deflate", self.L)
        return out[:n.value]

    def times(self):
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        self.L.crumble_gz_times(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"h2d_ms": a.value, "kernels_ms": b.value, "d2h_ms": c.value}


class StreamGroup(C.Structure):
    _fields_ = [("tid", C.c_int32), ("complete", C.c_int32), ("first_read", C.c_int64), ("n_reads", C.c_int64),
                ("n_bases", C.c_int64), ("byte_start", C.c_int64), ("byte_end", C.c_int64)]


class Stream:
    """crumble_stream (include/crumble_hip.h): a BAM file's records on the device from inflate to deflate"""

    def __init__(self, n_ref, header_bytes, device=0, library=None):
        L = self.L = _sig(library if library is not None else lib())
        vp, i64 = C.c_void_p, C.c_int64
        L.crumble_stream_create.restype = vp
        L.crumble_stream_create.argtypes = [C.c_int, C.c_int32, i64, C.POINTER(C.c_int)]
        L.crumble_stream_destroy.restype = None
        L.crumble_stream_destroy.argtypes = [vp]
        L.crumble_stream_feed.restype = C.c_int
        L.crumble_stream_feed.argtypes = [vp, vp, i64, vp, i64, C.c_int]
        L.crumble_stream_groups.restype = C.c_int
        L.crumble_stream_groups.argtypes = [vp, vp, C.c_int]
        L.crumble_stream_process.restype = C.c_int
        L.crumble_stream_process.argtypes = [vp, vp, C.c_int, C.c_int32, vp, vp]
        L.crumble_stream_drain.restype = C.c_int
        L.crumble_stream_drain.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, i64, C.POINTER(i64)]
        L.crumble_stream_peek.restype = C.c_int
        L.crumble_stream_peek.argtypes = [vp, i64, i64, vp, C.POINTER(i64), C.POINTER(i64), vp, i64]
        err = C.c_int(0)
        self.h = L.crumble_stream_create(device, n_ref, header_bytes, C.byref(err))
        if not self.h:
            raise CrumbleError(err.value, "crumble_stream_create", L)

    def reserve(self, comp_bytes, raw_bytes):
        self.L.crumble_stream_reserve.restype = C.c_int
        self.L.crumble_stream_reserve.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        rc = self.L.crumble_stream_reserve(self.h, comp_bytes, raw_bytes)
        if rc:
            raise CrumbleError(rc, "crumble_stream_reserve", self.L)

    def close(self):
        if self.h:
            self.L.crumble_stream_destroy(self.h)
            self.h = None

    def feed(self, comp, eof):
        a = np.frombuffer(comp, np.uint8) if not isinstance(comp, np.ndarray) else comp
        blocks, consumed, total = bgzf_scan(a, self.L) if a.size else (np.zeros(0, BLOCK_DTYPE), 0, 0)
        blocks = np.ascontiguousarray(blocks)
        rc = self.L.crumble_stream_feed(self.h, a.ctypes.data if a.size else None, consumed, blocks.ctypes.data if len(blocks) else None, len(blocks), 1 if eof else 0)
        if rc:
            raise CrumbleError(rc, "crumble_stream_feed", self.L)
        return consumed

    def groups(self):
        g = (StreamGroup * 4096)()
        n = self.L.crumble_stream_groups(self.h, g, 4096)
        return [g[i] for i in range(min(n, 4096))]

    def process(self, engine, i, bed=None):
        if bed is not None and len(bed):
            bs = np.ascontiguousarray(bed[:, 0], np.int32); be = np.ascontiguousarray(bed[:, 1], np.int32)
            rc = self.L.crumble_stream_process(self.h, engine.h, i, len(bs), bs.ctypes.data, be.ctypes.data)
        else:
            rc = self.L.crumble_stream_process(self.h, engine.h, i, 0, None, None)
        if rc:
            raise CrumbleError(rc, "crumble_stream_process", self.L)

    def pending(self):
        self.L.crumble_stream_pending.restype = C.c_int64
        self.L.crumble_stream_pending.argtypes = [C.c_void_p]
        return int(self.L.crumble_stream_pending(self.h))

    def drain(self, k, level=6, cap=None, flush=True):
        """the first k contig runs are done; BGZF members of what waits at the front of the buffer: whole members of 0xff00
        bytes as far as cap (bytes of the output buffer; default: enough for everything) allows, with flush the last, shorter
        one too when everything fits; the rest stays (pending())"""
        gs = self.groups()
        nbytes = gs[k - 1].byte_end if k else self.pending()
        if cap is None:
            cap = int(self.L.crumble_gz_deflate_bound(nbytes)) + 64
        out = np.empty(max(cap, 1), np.uint8)
        n = C.c_int64(0)
        rc = self.L.crumble_stream_drain(self.h, k, level, 1 if flush else 0, out.ctypes.data, cap, C.byref(n))
        if rc:
            raise CrumbleError(rc, "crumble_stream_drain", self.L)
        return out[:n.value]

    def drain_begin(self, k, level=6, flush=True, cap=None):
        """the drain's first half: marks the runs, enqueues the deflate of what a buffer of `cap` bytes takes; -> (record bytes
        taken, cap); drain_end(cap) collects the members"""
        gs = self.groups()
        nbytes = gs[k - 1].byte_end if k else self.pending()
        if cap is None:
            cap = int(self.L.crumble_gz_deflate_bound(nbytes)) + 64
        self.L.crumble_stream_drain_begin.restype = C.c_int
        self.L.crumble_stream_drain_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int64)]
        taken = C.c_int64(0)
        rc = self.L.crumble_stream_drain_begin(self.h, k, level, 1 if flush else 0, cap, C.byref(taken))
        if rc:
            raise CrumbleError(rc, "crumble_stream_drain_begin", self.L)
        return taken.value, cap

    def drain_end(self, cap):
        self.L.crumble_stream_drain_end.restype = C.c_int
        self.L.crumble_stream_drain_end.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        out = np.empty(max(cap, 1), np.uint8)
        n = C.c_int64(0)
        rc = self.L.crumble_stream_drain_end(self.h, out.ctypes.data, cap, C.byref(n))
        if rc:
            raise CrumbleError(rc, "crumble_stream_drain_end", self.L)
        return out[:n.value]

    def peek(self):
        """(buffer bytes, record offsets)"""
        ln, nr = C.c_int64(0), C.c_int64(0)
        self.L.crumble_stream_peek(self.h, 0, 0, None, C.byref(ln), C.byref(nr), None, 0)
        buf = np.empty(max(ln.value, 1), np.uint8)
        ro = np.empty(max(nr.value, 1), np.int64)
        rc = self.L.crumble_stream_peek(self.h, 0, ln.value, buf.ctypes.data, C.byref(ln), C.byref(nr), ro.ctypes.data, nr.value)
        if rc:
            raise CrumbleError(rc, "crumble_stream_peek", self.L)
        return buf[:ln.value], ro[:nr.value]
