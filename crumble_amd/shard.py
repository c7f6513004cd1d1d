"""Contig sharding across GPUs and the one collective of the path.

Contigs are the independent unit of crumble's column engine: every piece of
cross-column state is reset when the reference id changes
(/root/reference/snp_score.c:1478-1488), so contigs are packed onto ranks with
LPT (longest processing time first) and processed with no data-path exchange.
The only collective is the final sum of the statistics / quality-histogram
vector (int64[288]) -- torch.distributed all_reduce, i.e. RCCL over xGMI with
the "nccl" backend, gloo on CPU in the tests.
"""
import numpy as np

# GRCh38 primary contig lengths (chr1..22, X, Y): config 5 of BASELINE.json
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
          145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
          101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
          156040895, 57227415]


def lpt_pack(sizes, n_ranks):
    """Greedy LPT bin packing: returns (assignment list per rank, load per rank)."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    loads = [0] * n_ranks
    assign = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (loads[k], k))
        assign[r].append(i)
        loads[r] += sizes[i]
    for a in assign:
        a.sort()                       # keep header order inside a rank
    return assign, loads


def allreduce_stats(stats, dist=None, device=None):
    """Sum the int64 stats vector over all ranks (no-op without a process group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(stats, np.int64).copy()
    import torch
    t = torch.from_numpy(np.asarray(stats, np.int64).copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
