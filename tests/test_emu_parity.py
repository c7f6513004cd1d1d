"""Kernel logic vs the oracle WITHOUT a GPU: the literal kernel sources of
crumble_amd/csrc compiled for the CPU through tests/emu/hip_emu.h.  This is a
development aid for the no-GPU tier; the parity claim itself rests on the
-m gpu tests (tests/test_gpu_parity.py) that run the real HIP build."""
import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import HostBatch
import emu_lib
from parity_util import assert_parity, synth_batch, pblock_stress, pblock_long_stress


@pytest.mark.parametrize("level", [1, 5, 9])
def test_emu_chr20_small(product, level):
    hb = synth_batch("chr20", 0.0004)
    assert_parity(emu_lib.engine, ca.preset_params(level), hb)


def test_emu_ce(product):
    hb = synth_batch("ce", 0.3)
    assert_parity(emu_lib.engine, ca.preset_params(1), hb)


def test_emu_hifi_long_reads(product):
    """Long multi-op reads: the ballot-compacted record feed of k_columns<.,MULTI> and -y pbccs."""
    hb = synth_batch("hifi", 0.0002, read_len=3000)
    assert hb.cig_off[-1] > 3 * hb.n_reads
    assert_parity(emu_lib.engine, ca.preset_params(9, tech="pbccs"), hb)
    assert_parity(emu_lib.engine, ca.preset_params(5), hb)


@pytest.mark.parametrize("path", ["reads", "segments", "words"])
def test_emu_pblock_paths(product, path):
    """Both P-block implementations (one lane per read / segment-parallel) on short and long reads."""
    eng = emu_lib.with_options(pblock=path)
    for hb, p in ((synth_batch("chr20", 0.0004), ca.preset_params(9)),
                  (synth_batch("panel", 0.001), ca.preset_params(9, pblock=3, keep="30-31")),
                  (synth_batch("hifi", 0.0002, read_len=3000), ca.preset_params(9, tech="pbccs"))):
        assert_parity(eng, p, hb, check_columns=False)


@pytest.mark.parametrize("path", ["reads", "segments", "words"])
def test_emu_pblock_stress(product, path):
    """The pblock operator seam on raw quality strings: jumps, drifts, outliers, every chunk alignment."""
    pblock_stress(emu_lib.engine, path, n_reads=150)


@pytest.mark.parametrize("path", ["segments", "words"])
def test_emu_pblock_long_read_shapes(product, path):
    """Long-read-shaped quality strings through the two long-read decompositions (see pblock_long_stress)."""
    pblock_long_stress(emu_lib.engine, path, n_cases=25)


def test_emu_long_reads_deep_tiles(product):
    """Long multi-op reads at 150x: more records reach a 64-column tile than k_columns2 stages in one pass (KC2_R = 64),
    so a tile's slice of the hit list (k_p0_hitfill) is walked in several passes."""
    hb = synth_batch("hifi", 0.00004, contig_len=5000, read_len=1500, depth=100.0, err_indel=0.004)
    assert hb.cig_off[-1] > 3 * hb.n_reads and hb.n_bases > 80 * 5000
    assert_parity(emu_lib.with_options(kcol="v2"), ca.preset_params(9, tech="pbccs"), hb)


def test_emu_empty_and_passthrough(product):
    p = ca.preset_params(9)
    hb = HostBatch.from_reads([])
    assert_parity(emu_lib.engine, p, hb)
    reads = [dict(pos=10, cigar="*", seq="ACGTACGTAC", qual=[30, 31, 33, 36, 40, 12, 12, 13, 30, 30], flag=4),
             dict(pos=10, cigar="5S5I", seq="ACGTACGTAC", qual=35)]
    assert_parity(emu_lib.engine, p, HostBatch.from_reads(reads))
    assert_parity(emu_lib.engine, p, HostBatch.from_reads(reads, tid=-1))


def test_emu_consensus_modes(product):
    """-q (mode A), -Q (mode B) and both together (snp:1523-1543, 1593, 1624)."""
    hb = synth_batch("chr20", 0.0004)
    assert_parity(emu_lib.engine, ca.preset_params(5, min_qual_A=30, min_qual_B=0), hb)
    assert_parity(emu_lib.engine, ca.preset_params(5, min_qual_A=30, min_indel_A=60), hb)
    assert_parity(emu_lib.engine, ca.preset_params(9, min_qual_A=75, min_discrep_A=1.0, keep="25"), hb)
    assert_parity(emu_lib.engine, ca.preset_params(5, min_qual_A=0, min_qual_B=0), hb)


@pytest.mark.parametrize("kcol", ["v1", "v1perm", "v2"])
def test_emu_column_kernel_variants(product, kcol):
    """Every column kernel gives the same bits: k_columns with one block per base class, with per-column class
    numbering (class = idx(base) ^ idx(key base), undone by col_unpermute), and k_columns2 (pileup matrix staged in
    LDS, inline / parked split of the FP64 adds)."""
    eng = emu_lib.with_options(kcol=kcol)
    slow = kcol == "v2"        # k_columns2 under the emulator: a fiber rendezvous per matrix row (emu_lib.engine)
    hb = synth_batch("chr20", 0.0001 if slow else 0.0004)
    emu_lib.fin_counts()
    assert_parity(eng, ca.preset_params(5), hb)
    if slow:                   # k_columns2 finishes a tile one of two ways (cons_finalize_fast / the general path): both ran
        fast, general = emu_lib.fin_counts()
        assert fast > 10 and general > 3, (fast, general)
    if not slow:               # both consensus modes at once: k_columns whatever the option says
        assert_parity(eng, ca.preset_params(5, min_qual_A=30, min_indel_A=60), hb)
    if not slow:               # (mode A through k_columns2: test_emu_consensus_modes)
        assert_parity(eng, ca.preset_params(5, min_qual_A=30, min_qual_B=0), hb)
    assert_parity(eng, ca.preset_params(9, keep="37", perfect_col=1), hb)
    assert_parity(eng, ca.preset_params(9), synth_batch("panel", 0.001 if kcol != "v2" else 0.0002))
    assert_parity(eng, ca.preset_params(9, tech="pbccs"), synth_batch("hifi", 0.00007 if slow else 0.0002, read_len=3000))
    assert_parity(eng, ca.preset_params(1), synth_batch("ce", 0.08 if slow else 0.3))


def test_emu_indel_only(product):
    """indel_only.c:83-122 (D=20, Q=37) as a mode of the masking pass."""
    import oracle_lib as O
    from parity_util import cigar_zoo
    from test_pileup_cases import CASES
    for hb in (cigar_zoo(CASES, reps=10), synth_batch("chr20", 0.0004)):
        eng = emu_lib.engine(ca.preset_params(9))
        for d, q in ((20, 37), (3, 11)):
            got = eng.indel_only(hb, d, q)
            want = O.indel_only(hb, d, q)
            assert np.array_equal(got, want), (d, q, np.nonzero(got != want)[0][:10])
        eng.close()


def test_emu_upload_validation(product):
    """Errors are return codes, not asserts (the reference asserts at snp_score.c:1075)."""
    p = ca.preset_params(9)
    eng = emu_lib.engine(p)
    ok = [dict(pos=100, cigar="10M", seq="ACGTACGTAC", qual=30), dict(pos=105, cigar="10M", seq="ACGTACGTAC", qual=30)]
    eng.contig(HostBatch.from_reads(ok))
    with pytest.raises(ca.CrumbleError) as ei:      # unsorted
        eng.contig(HostBatch.from_reads(ok[::-1]))
    assert ei.value.code == -4
    with pytest.raises(ca.CrumbleError) as ei:      # quality >= 128
        eng.contig(HostBatch.from_reads([dict(pos=1, cigar="4M", seq="ACGT", qual=[1, 2, 200, 3])]))
    assert ei.value.code == -1
    with pytest.raises(ca.CrumbleError) as ei:      # CIGAR does not match l_qseq
        eng.contig(HostBatch.from_reads([dict(pos=1, cigar="5M", seq="ACGT", qual=9)]))
    assert ei.value.code == -1
    # an unsorted read that never enters the pileup is fine (FUNMAP mate)
    eng.contig(HostBatch.from_reads([ok[1], dict(pos=3, cigar="*", seq="ACGT", qual=9, flag=4)]))
    # offsets that do not start at 0 / run backwards / point past the arrays: an error code, never a wild read
    for field, idx, val in (("cig_off", 0, 1), ("seq_off", 0, 3), ("cig_off", 1, 7), ("seq_off", 1, -2), ("seq_off", 1, 10 ** 9)):
        hb = HostBatch.from_reads(ok)
        getattr(hb, field)[idx] = val
        with pytest.raises(ca.CrumbleError) as ei:
            eng.contig(hb)
        assert ei.value.code == -1, (field, idx, val)
    eng.close()


def _rejected_batches_leave_the_statistics(make_engine):
    """A batch that ends in CRUMBLE_EINVAL / CRUMBLE_EUNSORTED adds nothing to the engine's accumulated statistics vector
    (what crumble_hip_stats_allreduce sums), whichever way it is driven: crumble_hip_contig, crumble_hip_upload +
    crumble_hip_run (a quality >= 128 is seen by the replay, i.e. reported by the RUN: include/crumble_hip.h), or the
    five phases of a window."""
    from crumble_amd import windows as W
    p = ca.preset_params(5)
    eng = make_engine(p)
    ok = [dict(pos=100 + 3 * i, cigar="10M", seq="ACGTACGTAC", qual=30) for i in range(40)]
    hb_ok = HostBatch.from_reads(ok)
    badq = [dict(r) for r in ok]
    badq[17] = dict(badq[17], qual=[30, 30, 30, 200, 30, 30, 30, 30, 30, 30])
    hb_badq = HostBatch.from_reads(badq)
    hb_unsorted = HostBatch.from_reads(ok[:20] + ok[5:6] + ok[20:])
    eng.contig(hb_ok)
    base = eng.stats_read()
    assert base[ca.capi.ST_NAMES.index("reads")] == 40
    for hb, code in ((hb_badq, -1), (hb_unsorted, -4)):
        with pytest.raises(ca.CrumbleError) as ei:
            eng.contig(hb)
        assert ei.value.code == code
        assert np.array_equal(eng.stats_read(), base), "a rejected batch reached the accumulated statistics (contig)"
    # upload + run: the bad quality is the run's to report
    eng.upload(hb_badq)
    with pytest.raises(ca.CrumbleError) as ei:
        eng.run()
    assert ei.value.code == -1
    assert np.array_equal(eng.stats_read(), base), "a rejected batch reached the accumulated statistics (upload + run)"
    with pytest.raises(ca.CrumbleError) as ei:      # the unsorted batch is refused by the upload itself
        eng.upload(hb_unsorted)
    assert ei.value.code == -4
    # the phases of a window
    eng.upload(hb_badq)
    w = W.Window()
    w.col_start, w.col_end, w.stat_end, w.read_first, w.read_end = 100, 100 + 3 * 39 + 10, 100 + 3 * 39 + 10, 0, 40
    eng.set_window(w)
    cout = W.Carry()
    cin = W.Carry.start()
    eng.run_columns(); eng.run_depth(cin, cout); eng.run_triggers(); eng.run_fold(cin, cout)
    with pytest.raises(ca.CrumbleError) as ei:
        eng.run_finish()
    assert ei.value.code == -1
    assert np.array_equal(eng.stats_read(), base), "a rejected batch reached the accumulated statistics (window phases)"
    eng.contig(hb_ok)                               # still usable, and counting again
    assert np.array_equal(eng.stats_read(), 2 * base)
    eng.close()


def test_emu_rejected_batches_leave_the_statistics(product):
    _rejected_batches_leave_the_statistics(emu_lib.engine)


def test_emu_coded_download_contract(product):
    """crumble_hip_encode / _download_coded: nothing to encode before a run, a crumble_coded whose counts were changed since the
    encode is refused (the engine copies its own sizes, never the caller's), the plain download stays valid beside the coded
    one, a new run invalidates the coded result."""
    import ctypes as C
    p = ca.preset_params(9)
    hb = synth_batch("chr20", 0.0002)
    eng = emu_lib.engine(p)
    c = ca.Coded()
    assert eng.L.crumble_hip_encode(eng.h, C.byref(c)) == -1            # no run yet
    eng.upload(hb); eng.run()
    q, f, s = eng.download()
    c, f2, s2 = eng.download_coded()
    assert np.array_equal(eng.decode(c, hb.quals), q) and np.array_equal(f, f2) and np.array_equal(s, s2)
    q3, _, _ = eng.download()                                            # still there
    assert np.array_equal(q3, q)
    for field, val in (("n_escapes", c.n_escapes + 1), ("n_words", c.n_words - 1), ("n_blocks", c.n_blocks + 1), ("n_bases", c.n_bases - 16)):
        bad = ca.Coded.from_buffer_copy(c)
        setattr(bad, field, val)
        assert eng.L.crumble_hip_download_coded(eng.h, C.byref(bad), None, None) == -1, field
    eng.run()                                                            # a new run: encode again first
    assert eng.L.crumble_hip_download_coded(eng.h, C.byref(c), None, None) == -1
    c2, _, _ = eng.download_coded()
    assert np.array_equal(eng.decode(c2, hb.quals), q)
    eng.close()


def test_emu_rerun_and_engine_reuse(product):
    """run(); download == run(); run(); download (grow-only buffers, partially written arrays), and one engine
    reused for batches of different shapes gives what a fresh engine gives."""
    p = ca.preset_params(9)
    eng = emu_lib.engine(p)
    a, b = synth_batch("chr20", 0.0002), synth_batch("ce", 0.3)
    eng.upload(a); eng.run()
    q1, f1, s1 = eng.download()
    eng.run()
    q2, f2, s2 = eng.download()
    assert np.array_equal(q1, q2) and np.array_equal(f1, f2) and np.array_equal(s1, s2)
    for hb in (b, HostBatch.from_reads([]), a):
        got = eng.contig(hb)
        fresh = emu_lib.engine(p)
        want = fresh.contig(hb)
        fresh.close()
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
    eng.close()


def _expand_and_host_buffers(make_engine, library, hb, p):
    """crumble_hip_expand() re-runs the device-side pileup construction on the raw arrays an upload left on the device
    (what bench.py times per step): any number of times, same result; crumble_hip_host_alloc() buffers are ordinary
    destinations of a download; expand without a batch is an error, not a crash."""
    import oracle_lib as O
    eng = make_engine(p)
    want = O.contig(p, hb)
    eng.upload(hb); eng.run()
    got = eng.download()
    assert all(np.array_equal(x, y) for x, y in zip(got, want))
    eng.expand(); eng.run()
    assert all(np.array_equal(x, y) for x, y in zip(eng.download(), want))
    eng.expand(); eng.expand(); eng.run(); eng.run()
    qa, qp = ca.pinned_array(max(hb.n_bases, 1), np.uint8, library=library)
    fa, fp = ca.pinned_array(max(hb.n_reads, 1), np.uint8, library=library)
    try:
        qa[:] = 0xEE; fa[:] = 0xEE
        q, f, s = eng.download(out=(qa, fa))
        assert np.array_equal(q, want[0]) and np.array_equal(f, want[1]) and np.array_equal(s, want[2])
    finally:
        ca.pinned_free(qp, library=library); ca.pinned_free(fp, library=library)
    eng.close()
    fresh = make_engine(p)
    with pytest.raises(Exception):
        fresh.expand()
    # the copy half alone (crumble_hip_upload_raw), then expansion and run as separate calls: the streamed leg's split
    with pytest.raises(Exception):
        fresh.run()
    fresh.upload_raw(hb)
    with pytest.raises(Exception):
        fresh.run()                                     # raw arrays only: nothing to run yet
    fresh.expand(); fresh.run()
    assert all(np.array_equal(x, y) for x, y in zip(fresh.download(), want))
    assert fresh.upload_times()[0] >= 0.0
    fresh.close()


def test_emu_expand_and_host_buffers(product):
    _expand_and_host_buffers(emu_lib.engine, emu_lib.load(), synth_batch("chr20", 0.00012), ca.preset_params(5))


def test_emu_packed_bases(product):
    """crumble_batch.bases4 (BAM's two codes per byte, unpacked on the device by k_p0_unpack4) gives what the
    one-code-per-byte form gives: short and long reads, odd lengths, tiny and empty reads, the unmapped tail."""
    import copy
    p = ca.preset_params(9)
    L = emu_lib.load()
    tiny = HostBatch.from_reads([dict(pos=10, cigar="1M", seq="A", qual=[30]), dict(pos=10, cigar="*", seq="", qual=[]),
                                 dict(pos=11, cigar="3M", seq="ACG", qual=[30, 31, 2]), dict(pos=12, cigar="2M1I2M", seq="NACGT", qual=20),
                                 dict(pos=12, cigar="7M", seq="ACGTACG", qual=33)])
    for hb in (synth_batch("chr20", 0.0002), synth_batch("hifi", 0.0001, read_len=3001), synth_batch("ce", 0.2), tiny,
               synth_batch("chr20", 0.0035, read_len=36, depth=40.0)):      # > 65 536 reads: the offsets come from a device scan
        e = emu_lib.engine(p)
        want = e.contig(hb)
        hb4 = copy.copy(hb).pack4(L)
        assert hb4.struct().bases is None and hb4.struct().bases4
        got = e.contig(hb4)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
        up_off = e.transfer_bytes()[0]
        # ... and BAM's own per-read lengths (crumble_batch.l_qseq / n_cigar: the offsets are made on the device)
        hbl = copy.copy(hb4).lens()
        st = hbl.struct()
        assert st.l_qseq and st.n_cigar and not st.seq_off and not st.cig_off and st.n_bases == hb.n_bases
        got = e.contig(hbl)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
        up, down = e.transfer_bytes()
        # 6 bytes per read instead of 16 from 65 536 reads on (a smaller batch's offsets are made on the host: three launches saved)
        assert up == (up_off - 16 * (hb.n_reads + 1) + 6 * hb.n_reads if hb.n_reads > 65536 else up_off) and down >= hb.n_bases
        e.close()
    # a bad length is an error code like a bad offset (a negative l_qseq makes the device-made offsets run backwards)
    hbl = copy.copy(tiny).lens()
    hbl.l_qseq = hbl.l_qseq.copy(); hbl.l_qseq[2] = -3
    e = emu_lib.engine(p)
    with pytest.raises(ca.CrumbleError) as ei:
        e.contig(hbl)
    assert ei.value.code == -1
    # totals that do not match the lengths: refused on the host for a small batch, by the device scan's check for a large one
    # (the same bad batch must not behave differently by read count)
    for hb in (tiny, synth_batch("chr20", 0.0035, read_len=36, depth=40.0)):
        hbl = copy.copy(hb).lens()
        hbl.l_qseq = hbl.l_qseq.copy(); hbl.l_qseq[-1] -= 1          # n_bases now overstates the sum of the lengths
        with pytest.raises(ca.CrumbleError) as ei:
            e.contig(hbl)
        assert ei.value.code == -1 and "do not match" in str(ei.value)
    e.close()


def test_emu_coded_qualities(product):
    """crumble_batch.qualsp (2- or 4-bit codes into a table of at most sixteen qualities, made by crumble_pack_quals, expanded on
    the device by k_p0_quals) gives what the one-byte-per-quality form gives: four values (2 bits), eight and sixteen (4 bits),
    one value, lengths that are not a multiple of four / sixteen, an empty batch; seventeen values keep the bytes; the bytes
    that cross shrink accordingly; a table entry >= 128 is the same error as such a byte."""
    import copy
    L = emu_lib.load()
    rng = np.random.default_rng(5)
    def with_values(hb, vals):
        hb = copy.copy(hb)
        hb.quals = np.asarray(vals, np.uint8)[rng.integers(0, len(vals), hb.n_bases)]
        return hb
    base = synth_batch("chr20", 0.0002)
    tiny = HostBatch.from_reads([dict(pos=10, cigar="1M", seq="A", qual=[30]), dict(pos=10, cigar="*", seq="", qual=[]),
                                 dict(pos=11, cigar="3M", seq="ACG", qual=[30, 31, 2]), dict(pos=12, cigar="2M1I2M", seq="NACGT", qual=20),
                                 dict(pos=12, cigar="7M", seq="ACGTACG", qual=33)])
    empty = HostBatch.from_reads([])
    cases = [(base, 2), (with_values(base, [3, 9, 17, 22, 27, 33, 38, 41]), 4), (with_values(base, list(range(20, 36))), 4),
             (with_values(base, [30]), 2), (with_values(base, [0, 93]), 2), (tiny, 4), (empty, 2),
             (synth_batch("hifi", 0.0001, read_len=3001), 0), (with_values(base, list(range(20, 37))), 0)]
    for preset in (5, 9):
        p = ca.preset_params(preset)
        for hb, bits in cases:
            e = emu_lib.engine(p)
            want = e.contig(hb)
            up_plain = e.transfer_bytes()[0]
            hq = copy.copy(hb).packq(L)
            st = hq.struct()
            if bits == 0:
                assert getattr(hq, "qualsp", None) is None and st.quals and not st.qualsp
                e.close()
                continue
            assert st.quals is None and st.qualsp and st.qual_bits == bits, (st.qual_bits, bits)
            assert sorted(set(hq.qual_lut[:len(set(hb.quals.tolist()))].tolist())) == sorted(set(hb.quals.tolist()))
            got = e.contig(hq)
            assert all(np.array_equal(x, y) for x, y in zip(got, want))
            assert e.transfer_bytes()[0] == up_plain - hb.n_bases + (hb.n_bases * bits + 7) // 8
            # ... together with BAM's packed bases and per-read lengths: what a reader hands over
            got = e.contig(copy.copy(hq).pack4(L).lens())
            assert all(np.array_equal(x, y) for x, y in zip(got, want))
            # ... and the same engine goes back to the byte form (its quality array changes owner: arena view <-> own buffer)
            assert all(np.array_equal(x, y) for x, y in zip(e.contig(hb), want))
            assert all(np.array_equal(x, y) for x, y in zip(e.contig(hq), want))
            e.close()
    # a table entry >= 128: reported like a quality byte >= 128, by the run
    hq = copy.copy(base).packq(L)
    hq.qual_lut = hq.qual_lut.copy(); hq.qual_lut[1] = 200
    e = emu_lib.engine(ca.preset_params(5))
    with pytest.raises(ca.CrumbleError) as ei:
        e.contig(hq)
    assert ei.value.code == -1
    # a code width other than 2 / 4 is refused at upload
    hq = copy.copy(base).packq(L)
    hq.qual_bits = 3
    with pytest.raises(ca.CrumbleError):
        e.contig(hq)
    e.close()


def test_emu_staged_download(product):
    """crumble_hip_download through the page-locked staging lanes (8 host threads, double-buffered chunks): with the
    chunk shrunk so that this batch is hundreds of chunks -- whole chunks, a ragged last one, more lanes than chunks --
    the bytes equal the plain copy's."""
    p = ca.preset_params(5)
    hb = synth_batch("chr20", 0.0002)
    plain = emu_lib.engine(p)
    want = plain.contig(hb)
    plain.close()
    for chunk in (256, 4096, 1000003 // 256 * 256 + 256, hb.n_bases // 5 // 256 * 256 + 256):
        e = emu_lib.engine(p, dl_chunk=str(chunk))
        got = e.contig(hb)
        e.close()
        assert all(np.array_equal(x, y) for x, y in zip(got, want)), chunk


def test_emu_find_str_stress(product):
    """mask_LC_regions / find_STR seam on repeat-rich reads of many lengths vs the oracle (whose find_STR is
    pinned against the real reference str_finder.c)."""
    import oracle_lib as O
    from parity_util import str_stress_batch
    hb = str_stress_batch()
    p = ca.preset_params(1)
    eng = emu_lib.engine(p)
    eng.upload(hb)
    rng = np.random.default_rng(2)
    idx = rng.integers(0, hb.n_reads, 1500)
    lq = (hb.seq_off[idx + 1] - hb.seq_off[idx]).astype(np.int64)
    qpos1 = np.minimum((rng.random(len(idx)) * (lq + 1)).astype(np.int32) + 1, lq.astype(np.int32))
    kind = rng.integers(0, 2, len(idx)).astype(np.uint8)
    lo, hi = eng.mask_lc(idx, qpos1, kind)
    for i in range(len(idx)):
        assert O.mask_lc(p, hb, int(idx[i]), int(kind[i]), int(qpos1[i])) == (lo[i], hi[i]), (i, idx[i], qpos1[i])
    eng.close()


def test_emu_long_sparse_contig(product):
    """> 2^21 positions (more than 1024 scan chunks: the tile-to-tile carry of k_scan_partials, several
    prefix-max chunks) with a handful of read clusters, some with indels so that triggers exist."""
    rng = np.random.default_rng(5)
    reads = []
    for start in (100, 700_000, 1_400_000, 2_150_000, 2_200_000):
        pos = start
        for i in range(40):
            seq = "".join(rng.choice(list("ACGT"), 100))
            cigar = "100M" if i % 5 else "48M2D50M2S"
            if cigar != "100M":
                seq = seq[:98] + "AA"
            reads.append(dict(pos=pos, cigar=cigar, seq=seq, qual=rng.integers(5, 41, 100), mapq=int(rng.choice([0, 60]))))
            pos += int(rng.integers(0, 6))
    hb = HostBatch.from_reads(reads)
    for level in (5, 9):
        assert_parity(emu_lib.engine, ca.preset_params(level), hb)


def test_emu_run_paths(product):
    """crumble_hip_run's fast way (lists sized by guess, lengths on the device), the careful way (sized from a
    read-back) and the fast way with capacities too small for the batch (aborted by the scan, repeated carefully)
    give the oracle's bytes, flags, statistics and column records."""
    cases = ((ca.preset_params(5), synth_batch("chr20", 0.0002, err_indel=2e-3)),
             (ca.preset_params(9, tech="pbccs"), synth_batch("hifi", 0.00007, read_len=3000)),
             (ca.preset_params(1), synth_batch("ce", 0.1)))
    for i, (p, hb) in enumerate(cases):
        for opts in ({"run": "careful"}, {"run": "fast"}, {"run": "fast", "caps": 3}) + (({"run": "fast", "caps": 3, "kcol": "v2"},) if i == 0 else ()):
            assert_parity(emu_lib.with_options(**opts), p, hb)
    # the byte-wise replay ("replay" = flat: k_replay_flat, then k_restore's byte writes, k_readfix, k_keepfix) on a
    # long-read (multi-op) batch, a short-read one, and -S (k_replay_general) -- the paths the packed replay bypasses
    for p, hb in (cases[1], cases[0], (ca.preset_params(9, softclip=1), synth_batch("chr20", 0.0002, softclip_frac=0.3)),
                  (ca.preset_params(9, tech="pbccs", softclip=1), cases[1][1])):
        for opts in ({"replay": "flat"}, {"replay": "flat", "run": "fast", "caps": 3}):
            assert_parity(emu_lib.with_options(**opts), p, hb)
    # one engine: an aborted fast run, then batches that fit
    eng = emu_lib.engine(ca.preset_params(5), caps=2)
    import oracle_lib as O
    for hb in (cases[0][1], cases[2][1], cases[0][1]):
        q, f, s = eng.contig(hb)
        wq, wf, ws = O.contig(ca.preset_params(5), hb)
        assert np.array_equal(q, wq) and np.array_equal(f, wf) and np.array_equal(s, ws)
    eng.close()
