"""Loader + object wrapper for libcrumble_hip.so (the C ABI of include/crumble_hip.h).

There is deliberately no fallback here: if the HIP library or a GPU is missing
the calls raise.  The CPU oracle lives in oracle/ and is only imported by tests,
smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .capi import (ABI, Params, Batch, Column, Coded, DecodeCursor, HostBatch, COLUMN_DTYPE, ST_N, ROOT, HERE)

LIB_PATH = os.environ.get("CRUMBLE_HIP_LIB", os.path.join(HERE, "libcrumble_hip.so"))
_lib = None


class CrumbleError(RuntimeError):
    def __init__(self, code, where="", library=None):
        self.code = code
        L = library if library is not None else lib()
        msg = L.crumble_hip_strerror(code).decode()
        try:
            detail = L.crumble_hip_last_error().decode()
        except Exception:
            detail = ""
        super().__init__(f"{where}: {msg} ({code}) {detail}".strip())


def build_library(verbose=False):
    """make -C crumble_amd/csrc (hipcc --offload-arch=gfx950; works without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(HERE, "csrc")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode:
        print(r.stdout)
    if r.returncode:
        raise RuntimeError("building libcrumble_hip.so failed")
    return LIB_PATH


def load_library(path=None, make_default=True):
    global _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    L = C.CDLL(path, mode=C.RTLD_LOCAL)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    PP = C.POINTER(Params)
    sig = {
        "crumble_params_default": (None, [PP]),
        "crumble_params_preset": (C.c_int, [PP, C.c_int]),
        "crumble_params_tech": (C.c_int, [PP, C.c_char_p]),
        "crumble_params_keep_quals": (C.c_int, [PP, C.c_char_p, C.c_int]),
        "crumble_init_bins": (None, [PP, vp]),
        "crumble_bed_collapse": (C.c_int, [vp, vp, vp, C.c_int]),
        "crumble_hip_create": (vp, [PP, C.c_int, C.POINTER(C.c_int)]),
        "crumble_hip_destroy": (None, [vp]),
        "crumble_hip_set_option": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
        "crumble_hip_contig": (C.c_int, [vp, C.POINTER(Batch), vp, vp, vp]),
        "crumble_hip_contig_run": (C.c_int, [vp, C.POINTER(Batch)]),
        "crumble_hip_upload": (C.c_int, [vp, C.POINTER(Batch)]),
        "crumble_hip_upload_raw": (C.c_int, [vp, C.POINTER(Batch)]),
        "crumble_hip_expand": (C.c_int, [vp]),
        "crumble_hip_run": (C.c_int, [vp]),
        "crumble_hip_set_window": (C.c_int, [vp, vp]),
        "crumble_carry_init": (None, [vp]),
        "crumble_hip_run_columns": (C.c_int, [vp]),
        "crumble_hip_run_depth": (C.c_int, [vp, vp, vp]),
        "crumble_hip_run_triggers": (C.c_int, [vp]),
        "crumble_hip_run_fold": (C.c_int, [vp, vp, vp]),
        "crumble_hip_run_finish": (C.c_int, [vp]),
        "crumble_hip_download": (C.c_int, [vp, vp, vp, vp]),
        "crumble_hip_encode": (C.c_int, [vp, C.POINTER(Coded)]),
        "crumble_hip_download_coded": (C.c_int, [vp, C.POINTER(Coded), vp, vp]),
        "crumble_decode_seek": (None, [C.POINTER(Coded), i64, C.POINTER(DecodeCursor)]),
        "crumble_decode_run": (None, [C.POINTER(Coded), C.POINTER(DecodeCursor), i64, vp, vp]),
        "crumble_decode_quals": (None, [C.POINTER(Coded), vp, vp, C.c_int]),
        "crumble_hip_kernel_times": (C.c_int, [vp, vp, vp, vp, C.c_int]),
        "crumble_hip_upload_times": (C.c_int, [vp, vp, vp]),
        "crumble_hip_transfer_bytes": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "crumble_hip_stats_devptr": (vp, [vp]),
        "crumble_hip_stats_reset": (C.c_int, [vp]),
        "crumble_hip_stats_read": (C.c_int, [vp, vp]),
        "crumble_hip_stats_add": (C.c_int, [vp, vp]),
        "crumble_hip_stats_allreduce": (C.c_int, [vp, vp]),
        "crumble_hip_comm_init_all": (C.c_int, [C.c_int, vp, vp]),
        "crumble_hip_comm_destroy": (None, [vp]),
        "crumble_hip_columns": (i64, [vp, vp, i64]),
        "crumble_hip_mask_lc": (C.c_int, [vp, i64, vp, vp, vp, vp, vp]),
        "crumble_hip_pblock": (C.c_int, [vp, i64, vp, vp]),
        "crumble_hip_indel_only": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "crumble_hip_host_alloc": (vp, [C.c_size_t]),
        "crumble_hip_host_free": (None, [vp]),
        "crumble_hip_strerror": (C.c_char_p, [C.c_int]),
        "crumble_hip_last_error": (C.c_char_p, []),
        "crumble_hip_abi_version": (C.c_int, []),
        "crumble_hip_struct_size": (C.c_size_t, [C.c_int]),
        "crumble_set_host_threads": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        try:
            f = getattr(L, name)
        except AttributeError:        # tests/test_abi.py checks completeness
            continue
        f.restype, f.argtypes = res, args
    # the ABI handshake of include/crumble_hip.h: the structs declared in capi.py are the library's, field for field
    if hasattr(L, "crumble_hip_abi_version"):
        sizes = [C.sizeof(Batch), C.sizeof(Params), C.sizeof(Coded)]
        if L.crumble_hip_abi_version() != ABI or [L.crumble_hip_struct_size(i) for i in range(3)] != sizes:
            raise RuntimeError(f"{path}: built from another version of include/crumble_hip.h than crumble_amd/capi.py (ABI "
                               f"{L.crumble_hip_abi_version()} vs {ABI}; struct sizes "
                               f"{[L.crumble_hip_struct_size(i) for i in range(3)]} vs {sizes})")
    if make_default:
        _lib = L
    return L


def lib():
    return _lib if _lib is not None else load_library()


EXPORTED = [
    "crumble_params_default", "crumble_params_preset", "crumble_params_tech",
    "crumble_params_keep_quals", "crumble_init_bins", "crumble_bed_collapse",
    "crumble_hip_create", "crumble_hip_destroy", "crumble_hip_set_option", "crumble_hip_contig", "crumble_hip_contig_run",
    "crumble_hip_upload", "crumble_hip_upload_raw", "crumble_hip_expand", "crumble_hip_run", "crumble_hip_download",
    "crumble_pack_bases4", "crumble_pack_quals", "crumble_hip_encode", "crumble_hip_download_coded", "crumble_decode_seek", "crumble_decode_run", "crumble_decode_quals",
    "crumble_hip_set_window", "crumble_carry_init", "crumble_hip_run_columns", "crumble_hip_run_depth",
    "crumble_hip_run_triggers", "crumble_hip_run_fold", "crumble_hip_run_finish",
    "crumble_hip_kernel_times", "crumble_hip_upload_times", "crumble_hip_transfer_bytes", "crumble_hip_stats_devptr", "crumble_hip_stats_reset",
    "crumble_hip_stats_read", "crumble_hip_stats_add", "crumble_hip_stats_allreduce", "crumble_hip_comm_init_all",
    "crumble_hip_comm_destroy", "crumble_hip_columns",
    "crumble_hip_mask_lc", "crumble_hip_pblock", "crumble_hip_indel_only", "crumble_hip_host_alloc", "crumble_hip_host_free",
    "crumble_hip_strerror",
    "crumble_hip_last_error",
    "crumble_hip_abi_version", "crumble_hip_struct_size", "crumble_set_host_threads",
]


def pinned_array(n, dtype, library=None):
    """numpy array of n items in page-locked host memory (crumble_hip_host_alloc); keep the returned array alive and
    hand it to pinned_free() when done."""
    L = library if library is not None else lib()
    dt = np.dtype(dtype)
    nbytes = max(1, int(n) * dt.itemsize)
    p = L.crumble_hip_host_alloc(nbytes)
    if not p:
        raise MemoryError("crumble_hip_host_alloc")
    a = np.frombuffer((C.c_uint8 * nbytes).from_address(p), dtype=dt, count=int(n))
    return a, p


def pinned_free(p, library=None):
    (library if library is not None else lib()).crumble_hip_host_free(p)


def default_params():
    p = Params()
    lib().crumble_params_default(C.byref(p))
    return p


def preset_params(level=9, tech=None, keep=None, really_keep=None, **over):
    """crumble -<level> [-y tech] [-k list] [-K list] plus field overrides."""
    p = default_params()
    if level is not None:
        rc = lib().crumble_params_preset(C.byref(p), int(level))
        if rc:
            raise CrumbleError(rc, f"preset -{level}")
    if tech:
        lib().crumble_params_tech(C.byref(p), tech.encode())
    if keep:
        lib().crumble_params_keep_quals(C.byref(p), str(keep).encode(), 0)
    if really_keep:
        lib().crumble_params_keep_quals(C.byref(p), str(really_keep).encode(), 1)
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _vp(a):
    return a.ctypes.data if a is not None else None


class Engine:
    """One crumble_hip handle (one GPU, one stream)."""

    def __init__(self, params, device=0, library=None, options=None):
        """options: {"kcol": "v1"|"v1perm"|"v2", "pblock": "reads"|"segments", "replay": "flat"} -- implementation
        choices that never change a result (crumble_hip_set_option)."""
        err = C.c_int(0)
        self._p = params
        self.L = library if library is not None else lib()
        self.h = self.L.crumble_hip_create(C.byref(params), int(device), C.byref(err))
        if not self.h:
            raise CrumbleError(err.value, "crumble_hip_create", self.L)
        self._batch = None
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, name, value):
        self._chk(self.L.crumble_hip_set_option(self.h, str(name).encode(), str(value).encode()), f"set_option {name}={value}")

    def close(self):
        if getattr(self, "h", None):
            self.L.crumble_hip_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc, where):
        if rc:
            raise CrumbleError(rc, where, self.L)

    def upload(self, hb: HostBatch):
        self._batch = hb
        self._bstruct = hb.struct()
        self._chk(self.L.crumble_hip_upload(self.h, C.byref(self._bstruct)), "upload")

    def upload_raw(self, hb: HostBatch):
        """the copy half of upload() alone (crumble_hip_upload_raw): expand() + run() follow"""
        self._batch = hb
        self._bstruct = hb.struct()
        self._chk(self.L.crumble_hip_upload_raw(self.h, C.byref(self._bstruct)), "upload_raw")

    def expand(self):
        """the device half of upload() again, on the raw arrays already on the device"""
        self._chk(self.L.crumble_hip_expand(self.h), "expand")

    def run(self):
        self._chk(self.L.crumble_hip_run(self.h), "run")

    # one contig over several engines: windows (include/crumble_hip.h, crumble_amd/windows.py)
    def set_window(self, w):
        self._chk(self.L.crumble_hip_set_window(self.h, C.byref(w) if w is not None else None), "set_window")

    def run_columns(self):
        self._chk(self.L.crumble_hip_run_columns(self.h), "run_columns")

    def run_depth(self, carry_in, carry_out):
        self._chk(self.L.crumble_hip_run_depth(self.h, C.byref(carry_in), C.byref(carry_out)), "run_depth")

    def run_triggers(self):
        self._chk(self.L.crumble_hip_run_triggers(self.h), "run_triggers")

    def run_fold(self, carry_in, carry_out):
        self._chk(self.L.crumble_hip_run_fold(self.h, C.byref(carry_in), C.byref(carry_out)), "run_fold")

    def run_finish(self):
        self._chk(self.L.crumble_hip_run_finish(self.h), "run_finish")

    def download(self, want_flags=True, want_stats=True, out=None):
        """out: (qualities, flags) arrays of an earlier download to write into again (a caller that reuses its buffers,
        as crumble-hip's pool does, pays no first-touch page faults)."""
        hb = self._batch
        q = out[0] if out is not None else np.empty(max(hb.n_bases, 1), np.uint8)
        f = (out[1] if out is not None and out[1] is not None else np.zeros(max(hb.n_reads, 1), np.uint8)) if want_flags else None
        s = np.zeros(ST_N, np.int64) if want_stats else None
        self._chk(self.L.crumble_hip_download(self.h, _vp(q), _vp(f), _vp(s)), "download")
        return q[:hb.n_bases], (f[:hb.n_reads] if f is not None else None), s

    def download_coded(self, want_flags=True, want_stats=True, bufs=None):
        """crumble_hip_encode + crumble_hip_download_coded: (Coded, flags, stats).  The three streams land in numpy
        arrays kept alive on the returned struct (`_keep`); bufs = (codes, esc_off, esc) arrays of an earlier call (or
        page-locked ones) to write into again."""
        hb = self._batch
        c = Coded()
        self._chk(self.L.crumble_hip_encode(self.h, C.byref(c)), "encode")
        if bufs is not None and len(bufs[0]) >= c.n_words and len(bufs[1]) >= c.n_blocks and len(bufs[2]) >= c.n_escapes:
            codes, eo, esc = bufs
        else:
            codes = np.empty(max(1, c.n_words), np.uint32)
            eo = np.empty(max(1, c.n_blocks), np.int64)
            esc = np.empty(max(1, c.n_escapes), np.uint8)
        c.codes, c.esc_off, c.esc = codes.ctypes.data, eo.ctypes.data, esc.ctypes.data
        c._keep = (codes, eo, esc)
        f = np.zeros(max(hb.n_reads, 1), np.uint8) if want_flags else None
        s = np.zeros(ST_N, np.int64) if want_stats else None
        self._chk(self.L.crumble_hip_download_coded(self.h, C.byref(c), _vp(f), _vp(s)), "download_coded")
        return c, (f[:hb.n_reads] if f is not None else None), s

    def decode(self, c, quals_in, out=None, threads=0):
        """crumble_decode_quals: the coded result + the batch's input qualities -> the output qualities (host only)."""
        qin = np.ascontiguousarray(quals_in, np.uint8)
        q = out if out is not None else np.empty(max(1, c.n_bases), np.uint8)
        if c.n_bases:
            self.L.crumble_decode_quals(C.byref(c), qin.ctypes.data, q.ctypes.data, int(threads))
        return q[:c.n_bases]

    def contig(self, hb: HostBatch):
        """crumble_hip_contig: qualities, per-read flags, stats."""
        b = hb.struct()
        q = np.empty(max(hb.n_bases, 1), np.uint8)
        f = np.zeros(max(hb.n_reads, 1), np.uint8)
        s = np.zeros(ST_N, np.int64)
        self._batch = hb
        self._chk(self.L.crumble_hip_contig(self.h, C.byref(b), _vp(q), _vp(f), _vp(s)),
                  "contig")
        return q[:hb.n_bases], f[:hb.n_reads], s

    def columns(self):
        n = self.L.crumble_hip_columns(self.h, None, 0)
        if n < 0:
            raise CrumbleError(int(n), "columns")
        cols = np.zeros(max(n, 1), COLUMN_DTYPE)
        self.L.crumble_hip_columns(self.h, _vp(cols), n)
        return cols[:n]

    def kernel_times(self, maxk=64):
        names = (C.c_char_p * maxk)()
        ms = np.zeros(maxk, np.float32)
        by = np.zeros(maxk, np.float64)
        n = self.L.crumble_hip_kernel_times(self.h, C.cast(names, C.c_void_p), _vp(ms),
                                           _vp(by), maxk)
        return [(names[i].decode(), float(ms[i]), float(by[i])) for i in range(n)]

    def upload_times(self):
        """(h2d_ms, p0_ms) of the last upload(): raw copies, device-side expansion."""
        a, b = C.c_float(0), C.c_float(0)
        self._chk(self.L.crumble_hip_upload_times(self.h, C.byref(a), C.byref(b)), "upload_times")
        return float(a.value), float(b.value)

    def transfer_bytes(self):
        """(bytes the last upload copied from the host, bytes the last download copied back)"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.crumble_hip_transfer_bytes(self.h, C.byref(a), C.byref(b)), "transfer_bytes")
        return int(a.value), int(b.value)

    def mask_lc(self, read_idx, qpos1, is_indel):
        r = np.ascontiguousarray(read_idx, np.int64)
        q = np.ascontiguousarray(qpos1, np.int32)
        k = np.ascontiguousarray(is_indel, np.uint8)
        lo = np.zeros(len(r), np.int32)
        hi = np.zeros(len(r), np.int32)
        self._chk(self.L.crumble_hip_mask_lc(self.h, len(r), _vp(r), _vp(q), _vp(k),
                                            _vp(lo), _vp(hi)), "mask_lc")
        return lo, hi

    def pblock(self, seq_off, quals):
        so = np.ascontiguousarray(seq_off, np.int64)
        q = np.ascontiguousarray(quals, np.uint8).copy()
        self._chk(self.L.crumble_hip_pblock(self.h, len(so) - 1, _vp(so), _vp(q)), "pblock")
        return q

    def indel_only(self, hb, dist=20, q=37):
        """indel_only.c: keep +-dist bases around indels, everything else -> q."""
        self.upload(hb)
        out = np.empty(max(hb.n_bases, 1), np.uint8)
        self._chk(self.L.crumble_hip_indel_only(self.h, int(dist), int(q), _vp(out)), "indel_only")
        return out[:hb.n_bases]

    def stats_devptr(self):
        return self.L.crumble_hip_stats_devptr(self.h)

    def stats_reset(self):
        self._chk(self.L.crumble_hip_stats_reset(self.h), "stats_reset")

    def stats_add(self, stats):
        s = np.ascontiguousarray(stats, np.int64)
        assert len(s) == ST_N
        self._chk(self.L.crumble_hip_stats_add(self.h, _vp(s)), "stats_add")

    def stats_read(self):
        s = np.zeros(ST_N, np.int64)
        self._chk(self.L.crumble_hip_stats_read(self.h, _vp(s)), "stats_read")
        return s
