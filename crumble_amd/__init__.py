"""crumble_amd -- MI355X-native quality-quantisation engine (crumble's column engine).

The product is the C-ABI shared library ``libcrumble_hip.so`` built from
``crumble_amd/csrc`` (hand-written HIP kernels for gfx950 + plain-C host
layer, declared in ``include/crumble_hip.h``).  This package is the thin
Python driver used by the tests and by bench.py.
"""
from .capi import (Params, Batch, Column, Coded, DecodeCursor, HostBatch, COLUMN_DTYPE, ST_N, ST_HIST0,
                   ST_NAMES, RF_KEEP, RF_NOPBLOCK, RF_PASSTHRU, stats_dict, parse_cigar)
from . import shard
from .engine import (lib, load_library, build_library, default_params, preset_params,
                     Engine, CrumbleError, LIB_PATH, pinned_array, pinned_free)
