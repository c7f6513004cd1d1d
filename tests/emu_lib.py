"""Loads tests/emu/libcrumble_emu.so: the engine's literal kernel sources compiled
for the CPU through tests/emu/hip_emu.h.  TEST INFRASTRUCTURE ONLY -- lets the
no-GPU test tier exercise kernel logic; the -m gpu tier runs the real HIP build."""
import os
import subprocess

import crumble_amd
from crumble_amd.capi import ROOT

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libcrumble_emu.so")
_L = None


def load():
    global _L
    if _L is None:
        r = subprocess.run(["make", "-C", EMU_DIR], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            print(r.stdout)
            raise RuntimeError("building the emulated engine failed")
        _L = crumble_amd.load_library(EMU_LIB, make_default=False)
    return _L


def engine(params, **options):
    """Emulated engine.  k_columns2's wave-level idioms (a ballot or an LDS hand-off per matrix row) cost a fiber
    rendezvous each under emulation, ~10x the time of k_columns, so tests that are not about the column kernel get
    k_columns here unless they ask; the column-kernel, fuzz and tiny-read tests force every variant, and on the GPU
    every test runs the engine's own choice (k_columns2)."""
    options.setdefault("kcol", "v1")
    return crumble_amd.Engine(params, 0, library=load(), options=options)


def with_options(**options):
    """engine factory with crumble_hip_set_option choices applied (tests force every kernel variant)"""
    return lambda params: engine(params, **options)


def fin_counts():
    """(tiles k_columns2 finished through cons_finalize_fast, through the general finalize) since the last call"""
    import ctypes as C
    out = (C.c_ulonglong * 2)()
    load().crumble_emu_fin_counts(out)
    return int(out[0]), int(out[1])
