"""Parity tests proper: the real HIP build on a real MI355X, through the C ABI,
bit-exact against the CPU oracle (quality bytes, per-read flags, per-column
records, statistics)."""
import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import HostBatch
import oracle_lib as O
from parity_util import assert_parity, synth_batch, cigar_zoo, pblock_stress

pytestmark = pytest.mark.gpu


def gpu_engine(params):
    return ca.Engine(params, 0)


@pytest.mark.parametrize("level", [1, 3, 5, 7, 8, 9])
def test_chr20_presets(product, level):
    hb = synth_batch("chr20", 0.01)
    assert_parity(gpu_engine, ca.preset_params(level), hb)


@pytest.mark.parametrize("level", [1, 5])
def test_chr20_running_depth_epochs(product, level):
    """> 2^20 columns: the running depth average is halved (snp:1684-1687) inside the contig."""
    hb = synth_batch("chr20", 0.04)
    assert_parity(gpu_engine, ca.preset_params(level), hb)
    assert_parity(gpu_engine, ca.preset_params(level, over_depth=1.05), hb, check_columns=False)


def test_chr20_many_scan_chunks(product):
    """9.4 M columns: more than 1024 chunks in the position scan (the tile-to-tile carry of k_scan_partials),
    several halvings of the running depth average, thousands of prefix-max chunks."""
    hb = synth_batch("chr20", 0.15)
    assert hb.pos.max() - hb.pos.min() > 1024 * 8192
    assert_parity(gpu_engine, ca.preset_params(5), hb)


def test_ce_like(product):
    assert_parity(gpu_engine, ca.preset_params(1), synth_batch("ce", 1.0))


def test_panel_300x(product):
    assert_parity(gpu_engine, ca.preset_params(9), synth_batch("panel", 0.005))


def test_hifi_pbccs(product):
    hb = synth_batch("hifi", 0.002)
    hb.bed_start = np.arange(20000, 400000, 50000, dtype=np.int32)
    hb.bed_end = hb.bed_start + 2000
    assert_parity(gpu_engine, ca.preset_params(None, tech="pbccs"), hb)
    assert_parity(gpu_engine, ca.preset_params(5), hb)


@pytest.mark.parametrize("kcol", ["v1", "v1perm", "v2"])
def test_column_kernel_variants(product, kcol):
    """Every column kernel (k_columns with one block per base class / with per-column class numbering and the XOR
    un-permutation / k_columns2 with the LDS-staged pileup matrix) on every workload shape and consensus mode,
    whatever the engine would pick by itself."""
    def gpu_engine(q):
        return ca.Engine(q, 0, options={"kcol": kcol})
    hifi = synth_batch("hifi", 0.002)
    for hb, p in ((synth_batch("chr20", 0.01), ca.preset_params(5)),
                  (synth_batch("chr20", 0.002), ca.preset_params(5, min_qual_A=30, min_indel_A=60)),
                  (synth_batch("chr20", 0.002), ca.preset_params(5, min_qual_A=30, min_qual_B=0)),
                  (synth_batch("chr20", 0.002), ca.preset_params(9, keep="37", perfect_col=1)),
                  (synth_batch("ce", 1.0), ca.preset_params(1)),
                  (synth_batch("panel", 0.005), ca.preset_params(9)),
                  (hifi, ca.preset_params(None, tech="pbccs")),
                  (hifi, ca.preset_params(5, min_qual_A=30, min_indel_A=60))):
        assert_parity(gpu_engine, p, hb)
    from test_pileup_cases import CASES
    for level in (1, 9):
        assert_parity(gpu_engine, ca.preset_params(level, keep="30-33"), cigar_zoo(CASES))


@pytest.mark.parametrize("path", ["reads", "segments", "words"])
def test_pblock_paths(product, path):
    """Both P-block implementations (one lane per read / segment-parallel) on short and long reads."""
    def gpu_engine(q):
        return ca.Engine(q, 0, options={"pblock": path})
    for hb, p in ((synth_batch("chr20", 0.01), ca.preset_params(9)),
                  (synth_batch("panel", 0.005), ca.preset_params(9, pblock=3, keep="30-31")),
                  (synth_batch("hifi", 0.002), ca.preset_params(9, tech="pbccs")),
                  (synth_batch("hifi", 0.002), ca.preset_params(9, pblock=2))):
        assert_parity(gpu_engine, p, hb, check_columns=False)


@pytest.mark.parametrize("path", ["reads", "segments", "words"])
def test_pblock_stress(product, path):
    """The pblock operator seam on raw quality strings: jumps, drifts, outliers, every chunk alignment."""
    pblock_stress(gpu_engine, path, n_reads=2000)


@pytest.mark.parametrize("path", ["segments", "words"])
def test_pblock_long_read_shapes(product, path):
    """Long-read-shaped quality strings through the two long-read decompositions (see pblock_long_stress)."""
    from parity_util import pblock_long_stress
    pblock_long_stress(gpu_engine, path, seed=2, n_cases=150)


def test_long_reads_deep_tiles(product):
    """Long multi-op reads at 150x: more records reach a 64-column tile than k_columns2 stages in one pass (KC2_R = 64),
    so a tile's slice of the hit list (k_p0_hitfill) is walked in several passes."""
    hb = synth_batch("hifi", 0.0004, read_len=4000, depth=150.0, err_indel=0.002)
    assert hb.cig_off[-1] > 3 * hb.n_reads and hb.n_bases > 100 * hb.pos.max()
    assert_parity(gpu_engine, ca.preset_params(9, tech="pbccs"), hb)
    assert_parity(gpu_engine, ca.preset_params(5), hb)


def test_empty_and_passthrough(product):
    p = ca.preset_params(9)
    assert_parity(gpu_engine, p, HostBatch.from_reads([]))
    reads = [dict(pos=10, cigar="*", seq="ACGTACGTAC", qual=[30, 31, 33, 36, 40, 12, 12, 13, 30, 30], flag=4),
             dict(pos=10, cigar="5S5I", seq="ACGTACGTAC", qual=35)]
    assert_parity(gpu_engine, p, HostBatch.from_reads(reads))
    assert_parity(gpu_engine, p, HostBatch.from_reads(reads, tid=-1))


def test_mode_a_and_options(product):
    hb = synth_batch("chr20", 0.002)
    assert_parity(gpu_engine, ca.preset_params(5, min_qual_A=30, min_qual_B=0), hb)
    assert_parity(gpu_engine, ca.preset_params(5, min_qual_A=30, min_indel_A=60), hb)          # A and B together
    assert_parity(gpu_engine, ca.preset_params(9, min_qual_A=75, min_discrep_A=1.0, keep="25"), hb)
    assert_parity(gpu_engine, ca.preset_params(5, min_qual_A=0, min_qual_B=0), hb)            # no consensus at all
    assert_parity(gpu_engine, ca.preset_params(5, binary_qual=1, softclip=1), hb)
    assert_parity(gpu_engine, ca.preset_params(9, keep="37", perfect_col=1), hb)
    assert_parity(gpu_engine, ca.preset_params(3, really_keep="2", indel_fract=0.2), hb)


def test_mask_lc_and_pblock_seams(product):
    hb = synth_batch("chr20", 0.002)
    p = ca.preset_params(9)
    eng = gpu_engine(p)
    eng.upload(hb)
    rng = np.random.default_rng(1)
    idx = rng.integers(0, hb.n_reads, 300)
    idx = idx[(hb.flag[idx] & 4) == 0]
    lq = (hb.seq_off[idx + 1] - hb.seq_off[idx]).astype(np.int64)
    qpos1 = (rng.random(len(idx)) * lq).astype(np.int32) + 1
    kind = rng.integers(0, 2, len(idx)).astype(np.uint8)
    lo, hi = eng.mask_lc(idx, qpos1, kind)
    for i in range(len(idx)):
        assert O.mask_lc(p, hb, int(idx[i]), int(kind[i]), int(qpos1[i])) == (lo[i], hi[i])
    out = eng.pblock(hb.seq_off, hb.quals)
    for r in range(0, hb.n_reads, 37):
        a, b = hb.seq_off[r], hb.seq_off[r + 1]
        assert np.array_equal(out[a:b], O.pblock(hb.quals[a:b], p.pblock, p.qcap))
    eng.close()


def test_fp64_sqrt_div_match_host(product):
    """discrep = (m-c)/sqrt(m) must be the correctly rounded FP64 result."""
    hb = synth_batch("chr20", 0.002)
    p = ca.preset_params(5, min_discrep_B=0.05)     # make the comparison bite often
    assert_parity(gpu_engine, p, hb)


@pytest.mark.parametrize("level", [1, 5, 9])
def test_cigar_zoo(product, level):
    from test_pileup_cases import CASES
    hb = cigar_zoo(CASES)
    assert_parity(gpu_engine, ca.preset_params(level), hb)
    assert_parity(gpu_engine, ca.preset_params(level, keep="30-33", softclip=1), hb)
    assert_parity(gpu_engine, ca.preset_params(level, softclip=1, min_mqual=20, indel_fract=0.1), hb)


def test_large_properties(product):
    """Size-independent properties at a larger size than the oracle comparison:
    idempotent re-run, histogram total == bases, passthrough reads untouched at -5,
    every output quality is one of {qlow, qhigh} or an (optionally capped) input value."""
    hb = synth_batch("chr20", 0.1)
    p = ca.preset_params(5)
    eng = gpu_engine(p)
    q, f, s = eng.contig(hb)
    q2, f2, s2 = eng.contig(hb)
    assert np.array_equal(q, q2) and np.array_equal(f, f2) and np.array_equal(s, s2)
    assert s[ca.ST_HIST0:].sum() == hb.n_bases == s[ca.ST_NAMES.index("bases")]
    changed = q != hb.quals
    assert np.isin(q[changed], [p.qlow, p.qhigh, p.qcap]).all()
    pt = np.nonzero(f & ca.RF_PASSTHRU)[0]
    assert len(pt) > 0
    for r in pt[:200]:
        a, b = hb.seq_off[r], hb.seq_off[r + 1]
        assert np.array_equal(q[a:b], hb.quals[a:b])
    eng.close()


def test_packed_bases(product):
    """crumble_batch.bases4 (BAM's packing, unpacked by k_p0_unpack4) == the one-code-per-byte form == the oracle."""
    import copy
    import oracle_lib as O
    p = ca.preset_params(9)
    for hb in (synth_batch("chr20", 0.01), synth_batch("hifi", 0.002), synth_batch("panel", 0.005), synth_batch("ce", 0.3)):
        want = O.contig(p, hb)
        eng = gpu_engine(p)
        hb4 = copy.copy(hb).pack4(eng.L)
        got = eng.contig(hb4)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
        hbl = copy.copy(hb4).lens()                      # BAM's per-read lengths instead of the offset tables
        assert hbl.struct().l_qseq and not hbl.struct().seq_off
        got = eng.contig(hbl)
        eng.close()
        assert all(np.array_equal(x, y) for x, y in zip(got, want))


def test_staged_download(product):
    """crumble_hip_download of a batch large enough for the page-locked staging lanes (8 streams, double-buffered
    8 MiB chunks; include/crumble_hip.h "dl_chunk"): equal to the oracle, and to itself with other chunk sizes."""
    import oracle_lib as O
    hb = synth_batch("chr20", 0.05)
    assert hb.n_bases > 4 * (8 << 20)
    p = ca.preset_params(5)
    want = O.contig(p, hb)
    for chunk in (None, "1048576", "3000064"):
        eng = ca.Engine(p, 0, options={"dl_chunk": chunk} if chunk else None)
        got = eng.contig(hb)
        eng.close()
        assert all(np.array_equal(x, y) for x, y in zip(got, want)), chunk
    # a page-locked destination (crumble_hip_host_alloc): written by the GPU's own stores (k_d2h_store, "dl_kernel" = on, the
    # default) or by a copy-engine transfer (off) -- a length that is not a multiple of sixteen bytes, and the coded download's arrays
    qa, qp = ca.pinned_array(hb.n_bases, np.uint8)
    fa, fp = ca.pinned_array(hb.n_reads, np.uint8)
    ca_, cp = ca.pinned_array(hb.n_bases // 16 + 2, np.uint32)
    ea, ep = ca.pinned_array(hb.n_bases // 4096 + 2, np.int64)
    xa, xp = ca.pinned_array(hb.n_bases // 8 + 1, np.uint8)
    try:
        for mode in ("on", "off"):
            eng = ca.Engine(p, 0, options={"dl_kernel": mode})
            eng.upload(hb); eng.run()
            qa[:] = 0xEE; fa[:] = 0xEE
            q, f, s_ = eng.download(out=(qa, fa))
            assert np.array_equal(q, want[0]) and np.array_equal(f, want[1]) and np.array_equal(s_, want[2]), mode
            c_, f2, _ = eng.download_coded(want_stats=False, bufs=(ca_, ea, xa))
            assert np.array_equal(eng.decode(c_, hb.quals), want[0]) and np.array_equal(f2, want[1]), mode
            eng.close()
    finally:
        for p_ in (qp, fp, cp, ep, xp):
            ca.pinned_free(p_)


def test_rccl_stats_allreduce_single_rank(product):
    """crumble_hip_stats_allreduce: the one collective of the path, through RCCL (world size 1 here;
    the N>1 semantics are covered by tests/test_shard_gloo.py and bench.py --gpus N)."""
    import ctypes as C

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        pytest.skip("librccl.so not found")
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    hb = synth_batch("chr20", 0.002)
    eng = gpu_engine(ca.preset_params(5))
    _, _, s = eng.contig(hb)
    before = eng.stats_read()
    assert np.array_equal(before, s)
    rc = eng.L.crumble_hip_stats_allreduce(eng.h, comm)
    assert rc == 0, eng.L.crumble_hip_last_error()
    assert np.array_equal(eng.stats_read(), before)
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    eng.close()


def test_indel_only_mode(product):
    """indel_only.c:83-122 on the GPU vs the oracle."""
    from test_pileup_cases import CASES
    for hb in (cigar_zoo(CASES, reps=20), synth_batch("chr20", 0.01), synth_batch("hifi", 0.002)):
        eng = gpu_engine(ca.preset_params(9))
        for d, q in ((20, 37), (5, 2)):
            assert np.array_equal(eng.indel_only(hb, d, q), O.indel_only(hb, d, q))
        eng.close()


def test_too_deep_columns(product):
    """> MAX_DEPTH (20000) entries on a column: `goto too_deep` (snp:1493-1500) -- no consensus, no
    per-base work, no halving of the running depth average on that column (sequential k_epochs path)."""
    rng = np.random.default_rng(3)
    reads = []
    seqs = ["".join("ACGT"[i] for i in rng.integers(0, 4, 40)) for _ in range(8)]
    for i in range(300):                       # ordinary coverage before the pile
        reads.append(dict(pos=1000 + i, cigar="40M", seq=seqs[i % 8], qual=rng.integers(5, 41, 40).astype(np.uint8)))
    for i in range(20500):                     # the pile: every read covers 1400..1409
        p0 = 1380 + (i * 21) // 20500
        reads.append(dict(pos=p0, cigar="30M", seq=seqs[i % 8][:30], qual=(np.arange(30) % 30 + 8).astype(np.uint8),
                          mapq=int(rng.choice([0, 60]))))
    for i in range(300):
        reads.append(dict(pos=1401 + i, cigar="40M", seq=seqs[i % 8], qual=rng.integers(5, 41, 40).astype(np.uint8)))
    reads.sort(key=lambda r: r["pos"])
    hb = HostBatch.from_reads(reads)
    for level in (5, 9):
        q, f, s = assert_parity(gpu_engine, ca.preset_params(level), hb)
        assert s[ca.ST_NAMES.index("too_deep")] > 0


def test_upload_validation_errors(product):
    """Device-side input validation (p0_kernels.h): return codes instead of the reference's asserts."""
    eng = gpu_engine(ca.preset_params(9))
    ok = [dict(pos=100, cigar="10M", seq="ACGTACGTAC", qual=30), dict(pos=105, cigar="10M", seq="ACGTACGTAC", qual=30)]
    eng.contig(HostBatch.from_reads(ok))
    for reads, code in ((ok[::-1], -4),
                        ([dict(pos=1, cigar="4M", seq="ACGT", qual=[1, 2, 200, 3])], -1),
                        ([dict(pos=1, cigar="5M", seq="ACGT", qual=9)], -1)):
        with pytest.raises(ca.CrumbleError) as ei:
            eng.contig(HostBatch.from_reads(reads))
        assert ei.value.code == code
    eng.contig(HostBatch.from_reads(ok))            # the engine is still usable
    eng.close()


def test_rejected_batches_leave_the_statistics(product):
    """upload + run and the windowed phases report a bad quality through the run, and no rejected batch reaches the
    accumulated statistics vector"""
    from test_emu_parity import _rejected_batches_leave_the_statistics
    _rejected_batches_leave_the_statistics(gpu_engine)


@pytest.mark.parametrize("workload,ranks", [("chr20", 2), ("wgs", 2), ("wgs", 8)])
def test_bench_two_ranks(product, workload, ranks):
    """bench.py's N>1 control flow (LPT packing of the 24 contigs onto the ranks, barriers, max-over-ranks timing,
    stats / base-count all-reduce, one JSON line from rank 0) with 2 -- and, at the world size the target names, 8 --
    processes sharing this box's one GPU; gloo stands in for RCCL here."""
    import json, os, subprocess, sys
    from crumble_amd.capi import ROOT
    env = dict(os.environ, CRUMBLE_BENCH_BACKEND="gloo", CRUMBLE_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    scale = "0.01" if workload == "chr20" else "0.0005"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(29533 + ranks), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--workload", workload, "--scale", scale],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == ("weak" if workload == "chr20" else "strong")
    assert "roofline" in d and "cpu_baseline" not in d
    if workload == "wgs":
        assert d["config"]["contigs"] == 24 and 1.0 <= d["config"]["lpt_max_over_mean_load"] < 1.35
    else:
        # the driver's `--gpus N` line measures BASELINE config 5 beside the chr20 headline: 24 contigs LPT-packed over the
        # ranks (strong scaling), the statistics summed over the ranks, the summed histogram checked
        ws = d["wgs_strong"]
        assert "error" not in ws, ws
        assert ws["scaling"] == "strong" and ws["n_gpus"] == ranks and ws["value"] > 0 and ws["summed_histogram_ok"] is True
        assert sum(ws["contigs_per_rank"]) == 24 and 1.0 <= ws["lpt_max_over_mean_load"] < 1.35


@pytest.mark.parametrize("workload", ["chr20", "wgs"])
def test_bench_multi_gpu_rccl(product, workload):
    """The real thing when the box has it: one process per DISTINCT GPU, backend nccl (= RCCL over xGMI), the product's
    own crumble_hip_stats_allreduce.  Skipped on a one-GPU box (the gloo test above covers the control flow there)."""
    import json, os, subprocess, sys
    from crumble_amd.capi import ROOT
    from parity_util import gpu_count
    n = gpu_count()                                   # (asked in a child process: see parity_util.gpu_count)
    if n < 2:
        pytest.skip("one visible GPU")
    n = 2 if n < 4 else 4
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CRUMBLE_BENCH_BACKEND", None); env.pop("CRUMBLE_BENCH_DEVICE", None)
    scale = "0.02" if workload == "chr20" else "0.002"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                        "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                        "--gpus", str(n), "--steps", "2", "--warmup", "1", "--workload", workload, "--scale", scale],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == n and d["value"] > 0
    assert "RCCL" in d["config"]["stats_allreduce"]
    if workload == "chr20":
        assert d["wgs_strong"]["summed_histogram_ok"] is True and "crumble_hip_stats_allreduce" in d["wgs_strong"]["stats_allreduce"]
    if workload == "wgs":
        assert 1.0 <= d["config"]["lpt_max_over_mean_load"] < 1.5


def test_bench_other_configs_small(product):
    """bench.py's `other_configs` legs (configs 3, 4 and the streamed config 5) at small scales: every leg verifies
    the timed engine's own output against the oracle, and the streamed leg reports its overlap."""
    import bench
    import synth
    from crumble_amd import shard
    oc3 = bench.measure_config(ca, synth, "panel", 0.01, 0, steps=1, warmup=1, verify_seconds=1.0)
    assert oc3["verified_timed_output_leading_reads"] and oc3["pipeline_frac"] > 0
    oc4 = bench.measure_config(ca, synth, "hifi", 0.002, 0, steps=1, warmup=1, verify_seconds=1.0)
    assert oc4["verified_timed_output_leading_reads"] and "keep.bed" in oc4["workload"]
    w = bench.measure_wgs_streamed(ca, synth, shard, 0.0008, 0)
    assert w["verified_smallest_contig"] and 0 < w["overlap_wall_over_sum_of_stages"] <= 1.05
    assert set(w["stages_ms"]) == {"h2d", "p0_and_kernels", "d2h", "host_decode"}
    assert w["pcie_bytes_per_base"]["out"] < 0.5 < w["pcie_bytes_per_base"]["in"] < 2.0      # the coded download, the packed upload
    w = bench.measure_wgs_streamed(ca, synth, shard, 0.0008, 0, coded_in=True)                # ... and the qualities as codes, packed inside the clock
    assert w["verified_smallest_contig"] and set(w["stages_ms"]) == {"h2d", "p0_and_kernels", "d2h", "host_decode", "host_pack"}
    assert w["pcie_bytes_per_base"]["out"] < 0.5 < w["pcie_bytes_per_base"]["in"] < 1.0
    w = bench.measure_wgs_streamed(ca, synth, shard, 0.0008, 0, coded=False, coded_in=False)
    assert w["verified_smallest_contig"] and set(w["stages_ms"]) == {"h2d", "p0_and_kernels", "d2h"} and w["pcie_bytes_per_base"]["out"] >= 1.0
    assert 1.5 < w["pcie_bytes_per_base"]["in"] < 2.0                                          # one byte per quality


def test_coded_qualities_gpu(product):
    """crumble_batch.qualsp on the GPU (k_p0_quals<2> / <4>): the oracle's bytes, flags and statistics, as for the byte form."""
    import copy
    import oracle_lib as O
    rng = np.random.default_rng(11)
    a = synth_batch("chr20", 0.01)
    b = copy.copy(a)
    b.quals = np.array([3, 9, 17, 22, 27, 33, 38, 41, 12, 13], np.uint8)[rng.integers(0, 10, a.n_bases)]
    for preset, hb, bits in ((5, a, 2), (9, a, 2), (5, b, 4), (9, b, 4)):
        p = ca.preset_params(preset)
        want = O.contig(p, hb)
        hq = copy.copy(hb).pack4(ca.lib()).lens().packq(ca.lib())
        assert hq.qual_bits == bits and hq.struct().quals is None
        eng = gpu_engine(p)
        got = eng.contig(hq)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
        up = eng.transfer_bytes()[0]
        assert up < hb.n_bases * (0.5 + bits / 8.0 + 0.2)
        eng.close()


def test_find_str_stress(product):
    """mask_LC_regions / find_STR seam on repeat-rich reads of many lengths (see tests/test_emu_parity.py)."""
    from parity_util import str_stress_batch
    hb = str_stress_batch(n_reads=1500, seed=10)
    p = ca.preset_params(1)
    eng = gpu_engine(p)
    eng.upload(hb)
    rng = np.random.default_rng(3)
    idx = rng.integers(0, hb.n_reads, 20000)
    lq = (hb.seq_off[idx + 1] - hb.seq_off[idx]).astype(np.int64)
    qpos1 = np.minimum((rng.random(len(idx)) * (lq + 1)).astype(np.int32) + 1, lq.astype(np.int32))
    kind = rng.integers(0, 2, len(idx)).astype(np.uint8)
    lo, hi = eng.mask_lc(idx, qpos1, kind)
    for i in range(len(idx)):
        assert O.mask_lc(p, hb, int(idx[i]), int(kind[i]), int(qpos1[i])) == (lo[i], hi[i]), (i, idx[i], qpos1[i])
    eng.close()


def test_rerun_and_engine_reuse(product):
    """The work bench.py times: run(); download == run(); run(); download (grow-only buffers, partially written
    arrays), and one engine reused for batches of different shapes == the oracle every time."""
    p = ca.preset_params(9)
    eng = ca.Engine(p, 0)
    a, b, c = synth_batch("chr20", 0.01), synth_batch("panel", 0.005), synth_batch("hifi", 0.002)
    eng.upload(a); eng.run()
    q1, f1, s1 = eng.download()
    eng.run(); eng.run()
    q2, f2, s2 = eng.download()
    assert np.array_equal(q1, q2) and np.array_equal(f1, f2) and np.array_equal(s1, s2)
    import oracle_lib as O
    for hb in (b, c, a, HostBatch.from_reads([]), b):
        got = eng.contig(hb)
        want = O.contig(p, hb)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
    eng.close()


def test_expand_and_host_buffers(product):
    """crumble_hip_expand (the step bench.py times = expand + run) and crumble_hip_host_alloc through the HIP build"""
    from test_emu_parity import _expand_and_host_buffers
    _expand_and_host_buffers(lambda q: ca.Engine(q, 0), None, synth_batch("chr20", 0.01), ca.preset_params(5))
    _expand_and_host_buffers(lambda q: ca.Engine(q, 0), None, synth_batch("hifi", 0.002), ca.preset_params(9, tech="pbccs"))
