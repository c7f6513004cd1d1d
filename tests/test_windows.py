"""One contig cut into windows over several engines (include/crumble_hip.h "windows"): the owned
reads' quality bytes and flags and the summed statistics must equal the unsharded run's and the oracle's.  The CPU
tier runs the literal kernel sources under the emulator (both column kernels); the GPU tier adds sizes where the
running depth average halves several times inside the contig and the long-read path."""
import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import windows
import synth
import oracle_lib as O
from parity_util import gap_batch


def _check(make_engine, hb, params, n_windows, want=None):
    engines = [make_engine(params) for _ in range(n_windows)]
    try:
        q, f, s = windows.run(engines, hb, n_windows)
    finally:
        for e in engines:
            e.close()
    if want is None:
        want = O.contig(params, hb)
    wq, wf, ws = want
    assert np.array_equal(q, wq), ("quality bytes", int((q != wq).sum()), np.nonzero(q != wq)[0][:5])
    assert np.array_equal(f, wf), ("read flags", np.nonzero(f != wf)[0][:5])
    bad = np.nonzero(s != ws)[0]
    assert len(bad) == 0, ("statistics", [(ca.ST_NAMES[i] if i < len(ca.ST_NAMES) else i, int(s[i]), int(ws[i])) for i in bad[:6]])


def test_plan_covers_every_read_once():
    hb = synth.generate(synth.config("chr20", scale=0.0006, seed=3))
    for k in (1, 2, 3, 7):
        wins = windows.plan(hb, k)
        owned = []
        end = windows.ref_ends(hb)
        for lo, hi, w in wins:
            r0, r1 = lo + w.read_first, lo + w.read_end
            owned += list(range(r0, r1))
            assert w.col_start <= w.stat_end <= w.col_end
            assert (hb.pos[r0:r1] >= w.col_start).all() and (end[r0:r1] <= w.col_end).all()
            # every read that overlaps a decided column is in the slice
            ov = np.nonzero((end > w.col_start) & (hb.pos < w.col_end))[0]
            assert len(ov) == 0 or (ov.min() >= lo and ov.max() < hi)
        assert owned == list(range(hb.n_reads))
        for (lo, hi, w), (lo2, hi2, w2) in zip(wins, wins[1:]):
            assert w.stat_end == w2.col_start          # counted column ranges tile the contig: no column is dropped


@pytest.mark.parametrize("n_windows", [2, 5])
def test_emu_windows_read_spanning_coverage_gaps(n_windows):
    import emu_lib
    hb = gap_batch()
    wins = windows.plan(hb, n_windows)
    assert len(wins) == n_windows
    for (lo, hi, w), (lo2, hi2, w2) in zip(wins, wins[1:]):
        assert w.stat_end == w2.col_start and w.col_end >= w.stat_end
    p = ca.preset_params(5)
    want = O.contig(p, hb)
    assert want[2][ca.ST_NAMES.index("columns")] == 6000          # every column under the long deletion is counted
    _check(emu_lib.with_options(kcol="v1"), hb, p, n_windows, want=want)
    _check(emu_lib.with_options(kcol="v2"), hb, ca.preset_params(9), n_windows)


@pytest.mark.parametrize("kcol", ["v1", "v2"])
def test_emu_windows_chr20(kcol):
    import emu_lib
    hb = synth.generate(synth.config("chr20", scale=0.0005, seed=11))
    p = ca.preset_params(5)
    _check(emu_lib.with_options(kcol=kcol), hb, p, 3)


def test_emu_windows_panel_and_bed():
    import emu_lib
    cfg = synth.config("panel", scale=0.02, seed=5)
    hb0 = synth.generate(cfg)
    lo, hi = int(hb0.pos.min()), int(hb0.pos.max())
    bed = np.array([[lo + 200, lo + 900], [(lo + hi) // 2 - 300, (lo + hi) // 2 + 300]], np.int32)
    hb = synth.generate(cfg, bed=bed)
    p = ca.preset_params(9)
    _check(emu_lib.engine, hb, p, 4)


def test_emu_windows_more_windows_than_starts():
    import emu_lib
    hb = synth.generate(synth.config("chr20", scale=0.0001, seed=2))
    _check(emu_lib.engine, hb, ca.preset_params(9), 6)


def test_emu_window_phase_order_is_checked():
    import emu_lib
    hb = synth.generate(synth.config("chr20", scale=0.0001, seed=2))
    eng = emu_lib.engine(ca.preset_params(5))
    eng.upload(hb)
    with pytest.raises(ca.CrumbleError):
        eng.run_triggers()
    eng.run_columns()
    with pytest.raises(ca.CrumbleError):
        eng.run_fold(windows.Carry.start(), windows.Carry.start())
    bad = windows.Carry.start(); bad.total_col = -1
    with pytest.raises(ca.CrumbleError):
        eng.run_depth(bad, windows.Carry.start())
    # a carried state whose halving is overdue (total_col beyond 2^20 + 1: the columns at the halving point of the
    # window before were too deep, snp:1493-1500): the first counted column halves it (snp:1684-1687)
    late, out = windows.Carry.start(), windows.Carry.start()
    late.total_col, late.total_depth = (1 << 20) + 5, 31_000_000
    eng.set_window(windows.Window(int(hb.pos[0]), 2**31 - 2, 2**31 - 2, 0, 0, hb.n_reads))
    eng.run_columns()
    eng.run_depth(late, out)
    eng.run_triggers(); eng.run_fold(late, windows.Carry.start()); eng.run_finish()
    cols = eng.columns()
    d = cols["n_plp"][cols["skipped"] != 1].astype(np.int64)         # counted columns (all-refskip ones are not)
    assert out.total_col == ((late.total_col + 1) >> 1) + len(d) - 1
    assert out.total_depth == ((late.total_depth + int(d[0])) >> 1) + int(d[1:].sum())
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,scale,preset,nw", [("chr20", 0.08, 5, 4), ("chr20", 0.02, 9, 3), ("panel", 1.0, 9, 5)])
def test_gpu_windows_match_the_whole_contig(product, name, scale, preset, nw):
    """chr20 at scale 0.08: 5 M columns, the depth average halves eight times; cuts fall inside live STR windows."""
    hb = synth.generate(synth.config(name, scale=scale, seed=21))
    p = ca.preset_params(preset)
    eng = ca.Engine(p, 0)
    want = eng.contig(hb)
    eng.close()
    _check(lambda pp: ca.Engine(pp, 0), hb, p, nw, want=want)


@pytest.mark.gpu
def test_gpu_windows_vs_oracle_and_long_reads(product):
    hb = synth.generate(synth.config("chr20", scale=0.004, seed=8))
    _check(lambda pp: ca.Engine(pp, 0), hb, ca.preset_params(5), 3)
    hf = synth.generate(synth.config("hifi", scale=0.004, seed=9))
    p = ca.preset_params(None, tech="pbccs")
    eng = ca.Engine(p, 0)
    want = eng.contig(hf)
    eng.close()
    _check(lambda pp: ca.Engine(pp, 0), hf, p, 3, want=want)


def _fuzz_windows(make_engine, seeds, max_bases):
    from fuzz_util import random_case
    for seed in seeds:
        p, hb, desc = random_case(seed, max_bases=max_bases)
        if hb.n_reads < 8 or hb.tid < 0:
            continue
        try:
            _check(make_engine, hb, p, 2 + seed % 4)
        except AssertionError as e:
            raise AssertionError(f"{desc}: {e}") from e


@pytest.mark.parametrize("seed", range(300, 306))
def test_fuzz_windows_emulated(product, seed):
    """Random option sets x random alignments (long reads, BEDs, -k/-K, -S, deep spots) cut into 2..5 windows."""
    import emu_lib
    _fuzz_windows(emu_lib.with_options(kcol=("v1", "v2")[seed & 1]), [seed], 120_000)


def test_tiny_reads_windows_emulated(product):
    """Reads of 0..40 bases with clips, indels, skips and unmapped records: cuts between reads that start within a
    few bases of each other (several windows' halos overlap)."""
    import emu_lib
    from test_fuzz_parity import tiny_reads_batch
    hb = tiny_reads_batch(5)
    for p in (ca.preset_params(9), ca.preset_params(5, pblock=2, keep="30"), ca.preset_params(9, softclip=1, binary_qual=1)):
        _check(emu_lib.engine, hb, p, 5)


@pytest.mark.gpu
def test_fuzz_windows_gpu(product):
    _fuzz_windows(lambda pp: ca.Engine(pp, 0), range(2000, 2060), 400_000)


def test_emu_window_then_whole_batch_again(product):
    """crumble_hip_set_window(NULL) after a windowed run: the same upload gives the whole contig's result again
    (and a window's statistics only count its own columns and reads)."""
    import emu_lib
    hb = synth.generate(synth.config("chr20", scale=0.0003, seed=4))
    p = ca.preset_params(5)
    eng = emu_lib.engine(p)
    want = eng.contig(hb)
    (lo0, hi0, w0), (lo1, hi1, w1) = windows.plan(hb, 2)
    assert lo0 == 0
    eng.upload(windows._slice(hb, lo0, hi0))
    eng.set_window(w0)
    c, n = windows.Carry.start(), windows.Carry.start()
    eng.run_columns(); eng.run_depth(c, n); eng.run_triggers(); eng.run_fold(c, n); eng.run_finish()
    q0, f0, s0 = eng.download()
    r1 = int(w0.read_end)
    assert np.array_equal(q0[:int(hb.seq_off[r1])], want[0][:int(hb.seq_off[r1])]) and np.array_equal(f0[:r1], want[1][:r1])
    assert s0[ca.ST_NAMES.index("reads")] == r1 and 0 < s0[ca.ST_NAMES.index("columns")] < want[2][ca.ST_NAMES.index("columns")]
    eng.upload(hb)
    eng.set_window(w0)                       # a window on the whole batch ...
    eng.set_window(None)                     # ... taken back
    eng.run()
    got = eng.download()
    eng.close()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_emu_windows_contig_starting_at_position_zero(product):
    """The start-of-contig carry has max_pos2 == 0: position 0 is not "in window" for that (snp:1880 tests
    min_pos != INT_MAX, not pos <= max_pos2) -- found by tools/gpu_fuzz.py seed 5121."""
    import emu_lib
    hb = synth.generate(synth.config("chr20", scale=0.0003, seed=12))
    hb.pos -= hb.pos.min()
    assert hb.pos[0] == 0
    _check(emu_lib.engine, hb, ca.preset_params(5), 3)
    _check(emu_lib.engine, hb, ca.preset_params(9, qlow=5), 2)
