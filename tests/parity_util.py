"""Shared helpers for the parity tests (HIP engine or emulated engine vs oracle)."""
import numpy as np

import crumble_amd as ca
from crumble_amd import synth
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
    finally:
        eng.close()
    return q, f, s
