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
    eng = emu_lib.engine(p, kcol=("v1", "v2")[rank & 1])   # one engine per rank, reused for all its contigs
    ok = True
    for tid in assign[rank]:
        cfg = synth.config("chr20", scale=scales[tid] * 0.001, seed=100 + tid, tid=tid)
        hb = synth.generate(cfg, threads=2)
        q1, f1, _ = eng.contig(hb)
        wq, wf, _ = O.contig(p, hb)
        ok = ok and np.array_equal(q1, wq) and np.array_equal(f1, wf)
    local = eng.stats_read()                                # what crumble_hip_stats_allreduce sums on the GPU
    eng.close()
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
