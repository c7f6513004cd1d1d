"""Python driver for the C synthetic-alignment generator (tests/synth/crumble_synth.h): test / bench
infrastructure in its own libcrumble_synth.so -- the product libraries do not contain it."""
import ctypes as C
import os
import subprocess

import numpy as np

from crumble_amd.capi import Batch, HostBatch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcrumble_synth.so")
_L = None


def lib():
    global _L
    if _L is None:
        r = subprocess.run(["make", "-C", HERE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            print(r.stdout)
            raise RuntimeError("building libcrumble_synth.so failed")
        _L = C.CDLL(LIB, mode=C.RTLD_LOCAL)
    return _L


class SynthCfg(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("tid", C.c_int32), ("contig_len", C.c_int32),
        ("read_len", C.c_int32), ("read_len_sigma", C.c_int32),
        ("depth", C.c_double), ("depth_sigma", C.c_double), ("qual_model", C.c_int32),
        ("het_snp", C.c_double), ("hom_snp", C.c_double), ("indel_var", C.c_double),
        ("err_indel", C.c_double), ("str_every", C.c_int32),
        ("mapq0_frac", C.c_double), ("mapq_mid_frac", C.c_double),
        ("softclip_frac", C.c_double), ("unmapped_frac", C.c_double),
        ("refskip_frac", C.c_double), ("target_len", C.c_int32),
        ("target_every", C.c_int32), ("spike_pos", C.c_int32), ("spike_depth", C.c_int32),
    ]


def _setup():
    L = lib()
    L.crumble_synth_config.restype = C.c_int
    L.crumble_synth_config.argtypes = [C.POINTER(SynthCfg), C.c_char_p, C.c_double]
    L.crumble_synth_generate.restype = C.c_void_p
    L.crumble_synth_generate.argtypes = [C.POINTER(SynthCfg), C.c_int]
    L.crumble_synth_batch.restype = C.POINTER(Batch)
    L.crumble_synth_batch.argtypes = [C.c_void_p]
    L.crumble_synth_n_bases.restype = C.c_int64
    L.crumble_synth_n_bases.argtypes = [C.c_void_p]
    L.crumble_synth_free.restype = None
    L.crumble_synth_free.argtypes = [C.c_void_p]
    return L


def config(name, scale=1.0, **over):
    L = _setup()
    c = SynthCfg()
    rc = L.crumble_synth_config(C.byref(c), name.encode(), float(scale))
    if rc:
        raise ValueError(f"unknown synthetic config {name!r}")
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def _view(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dt)
    buf = (C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dt, n).copy()


def generate(cfg, threads=0, bed=None):
    """Runs the generator and copies the result into a HostBatch."""
    L = _setup()
    s = L.crumble_synth_generate(C.byref(cfg), int(threads))
    if not s:
        raise MemoryError("crumble_synth_generate")
    try:
        b = L.crumble_synth_batch(s).contents
        n = int(b.n_reads)
        nb = int(L.crumble_synth_n_bases(s))
        seq_off = _view(b.seq_off, n + 1, np.int64)
        cig_off = _view(b.cig_off, n + 1, np.int64)
        hb = HostBatch(b.tid, _view(b.pos, n, np.int32), _view(b.mapq, n, np.uint8),
                       _view(b.flag, n, np.uint16), seq_off, cig_off,
                       _view(b.cigar, int(cig_off[-1]) if n else 0, np.uint32),
                       _view(b.bases, nb, np.uint8), _view(b.quals, nb, np.uint8), bed)
    finally:
        L.crumble_synth_free(s)
    return hb
