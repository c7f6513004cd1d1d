"""N>1 path on CPU: world_size-2 gloo.  Contigs are LPT-packed onto ranks, each rank runs its contigs through the
ENGINE (the literal kernel sources under the CPU emulator, tests/emu -- same C ABI, same host code as the GPU
build), the per-rank engine's accumulated statistics vector is all-reduced; every contig's bytes and flags must
equal the oracle's and the reduced vector the single-process oracle sum."""
import os
import socket
import sys

import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import shard
import synth


def test_lpt_pack_balances_grch38():
    for n in (1, 2, 4, 8):
        assign, loads = shard.lpt_pack(shard.GRCH38, n)
        assert sorted(i for a in assign for i in a) == list(range(24))
        assert max(loads) <= sum(shard.GRCH38) / n * 1.06, (n, loads)
    assign, loads = shard.lpt_pack([5, 4, 3, 3, 1], 2)
    assert sorted(loads) == [8, 8]


def _worker(rank, world, port, scales, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    import oracle_lib as O
    import emu_lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [int(1e6 * s) for s in scales]
    assign, _ = shard.lpt_pack(sizes, world)
    p = ca.preset_params(5)
    # rank 0: one engine reused for all its contigs; rank 1: an engine per contig (several engines on one device), their
    # vectors folded into the first with crumble_hip_stats_add before the rank's one all-reduce (bench.py's wgs_strong),
    # the batches in the form crumble-hip sends them (packed bases, per-read lengths), the result taken in coded form
    engs = [emu_lib.engine(p, kcol=("v1", "v2")[rank & 1])]
    ok = True
    for k, tid in enumerate(assign[rank]):
        cfg = synth.config("chr20", scale=scales[tid] * 0.001, seed=100 + tid, tid=tid)
        hb = synth.generate(cfg, threads=2)
        if rank == 1 and k > 0:
            engs.append(emu_lib.engine(p, kcol="v1"))
        eng = engs[-1]
        if rank == 1:
            import copy
            eng.upload(copy.copy(hb).pack4(emu_lib.load()).lens()); eng.run()
            c, f1, _ = eng.download_coded(want_stats=False)
            q1 = eng.decode(c, hb.quals)
        else:
            q1, f1, _ = eng.contig(hb)
        wq, wf, _ = O.contig(p, hb)
        ok = ok and np.array_equal(q1, wq) and np.array_equal(f1, wf)
    if len(engs) > 1:
        engs[0].stats_add(sum(e.stats_read() for e in engs[1:]))
    local = engs[0].stats_read()                            # what crumble_hip_stats_allreduce sums on the GPU
    for e in engs:
        e.close()
    assert ok, "engine output differs from the oracle"
    total = shard.allreduce_stats(local, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, total))


def test_world_size_2_stats_allreduce(product):
    import torch.multiprocessing as mp
    import oracle_lib as O
    scales = [1.0, 0.6, 0.5, 0.3]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scales, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    # single-process reference
    p = ca.preset_params(5)
    want = np.zeros(ca.ST_N, np.int64)
    for tid, sc in enumerate(scales):
        hb = synth.generate(synth.config("chr20", scale=sc * 0.001, seed=100 + tid, tid=tid), threads=2)
        want += O.contig(p, hb)[2]
    assert np.array_equal(res[0], want) and np.array_equal(res[1], want)
    assert want[ca.ST_HIST0:].sum() == want[ca.ST_NAMES.index("bases")]


def _window_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    import emu_lib
    from crumble_amd import windows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hb = synth.generate(synth.config("chr20", scale=0.0006, seed=41), threads=2)     # every rank sees the contig
    p = ca.preset_params(5)
    eng = emu_lib.engine(p, kcol=("v1", "v2")[rank & 1])
    lo, hi, w, wq, wf, ws = windows.run_rank(eng, hb, dist)
    total = shard.allreduce_stats(eng.stats_read(), dist)
    eng.close()
    r0, r1 = lo + int(w.read_first), lo + int(w.read_end)
    b0 = int(hb.seq_off[r0]) - int(hb.seq_off[lo])
    own_q = wq[b0:b0 + int(hb.seq_off[r1] - hb.seq_off[r0])]
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, r0, r1, own_q, wf[int(w.read_first):int(w.read_end)], total))


def test_world_size_2_one_contig_in_windows(product):
    """One contig cut into two windows, one process (and one emulated engine) per window: the carried depth average
    and STR window state travel rank to rank (crumble_amd/windows.py run_rank); owned bytes + flags of both ranks
    put together and the all-reduced statistics equal the oracle's run of the whole contig."""
    import torch.multiprocessing as mp
    import oracle_lib as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_window_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    hb = synth.generate(synth.config("chr20", scale=0.0006, seed=41), threads=2)
    wq, wf, ws = O.contig(ca.preset_params(5), hb)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == hb.n_reads and res[0][2] > 0
    for rank, r0, r1, oq, of, total in res:
        assert np.array_equal(oq, wq[int(hb.seq_off[r0]):int(hb.seq_off[r1])]), rank
        assert np.array_equal(of, wf[r0:r1]), rank
        assert np.array_equal(total, ws), rank
