"""One contig over several engines (include/crumble_hip.h "windows").

plan()   cuts a contig's reads (file order) into windows of about equal bases: window k owns the reads that start in
         [w_k, w_k+1) and is uploaded with the reads around them that overlap the columns it decides.
run()    the five-phase protocol over one engine per window: columns, triggers and the finish of all windows are
         independent (one device each); the two small chained phases hand the running depth average and the STR
         window state from window to window.  Returns what Engine.contig() returns for the whole contig.

The same schedule in C: crumble_amd/csrc/cli.c (crumble-hip -j N -w).
"""
import ctypes as C

import numpy as np

from .capi import HostBatch
from .engine import ST_N


class Window(C.Structure):
    """crumble_window."""
    _fields_ = [("col_start", C.c_int32), ("col_end", C.c_int32), ("stat_end", C.c_int32), ("pad_", C.c_int32),
                ("read_first", C.c_int64), ("read_end", C.c_int64)]


class Carry(C.Structure):
    """crumble_carry."""
    _fields_ = [("total_col", C.c_int64), ("total_depth", C.c_int64),
                ("min_pos", C.c_int32), ("max_pos", C.c_int32), ("min_pos2", C.c_int32), ("max_pos2", C.c_int32)]

    @classmethod
    def start(cls):
        """The state at the start of a contig (snp_score.c:1478-1488)."""
        return cls(0, 0, 2**31 - 1, 0, 2**31 - 1, 0)

    def tuple(self):
        return (self.total_col, self.total_depth, self.min_pos, self.max_pos, self.min_pos2, self.max_pos2)


def ref_ends(hb):
    """pos + reference length of every read (M D N = X consume the reference)."""
    ops = hb.cigar
    cons = np.isin(ops & 15, (0, 2, 3, 7, 8)) * (ops >> 4).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(cons)])
    return hb.pos.astype(np.int64) + (csum[hb.cig_off[1:]] - csum[hb.cig_off[:-1]])


def _slice(hb, lo, hi):
    b0, c0 = int(hb.seq_off[lo]), int(hb.cig_off[lo])
    b1, c1 = int(hb.seq_off[hi]), int(hb.cig_off[hi])
    w = HostBatch(hb.tid, hb.pos[lo:hi], hb.mapq[lo:hi], hb.flag[lo:hi], hb.seq_off[lo:hi + 1] - b0,
                  hb.cig_off[lo:hi + 1] - c0, hb.cigar[c0:c1], hb.bases[b0:b1], hb.quals[b0:b1])
    w.bed_start, w.bed_end = hb.bed_start, hb.bed_end
    return w


def plan(hb, n_windows):
    """[(batch slice lo, hi, Window)] -- windows of about equal bases, cut between reads of different start."""
    n = hb.n_reads
    if n == 0:
        return []
    end = ref_ends(hb)
    pmax = np.maximum.accumulate(end)
    pos = hb.pos.astype(np.int64)
    firsts = [0]
    for k in range(1, n_windows):
        r = int(np.searchsorted(hb.seq_off, hb.n_bases * k // n_windows, side="left"))
        r = min(max(r, 0), n - 1)
        r = int(np.searchsorted(pos, pos[r], side="left"))     # all reads of one start position go together
        if r > firsts[-1]:
            firsts.append(r)
    firsts.append(n)
    out = []
    for k in range(len(firsts) - 1):
        r0, r1 = firsts[k], firsts[k + 1]
        w0 = int(pos[r0])
        e_k = max(int(end[r0:r1].max()), w0)
        # the window decides AND counts through to the next window's start: columns between the end of the owned
        # reads and w_k+1 can still be covered by a read of an earlier window (a long deletion / splice over a
        # coverage gap); the lo rule below already has every such read in the slice
        stat_end = int(pos[r1]) if r1 < n else max(e_k, int(pmax[-1]))
        e_k = max(e_k, stat_end)
        lo = min(int(np.searchsorted(pmax, w0, side="right")), r0) if k else 0
        hi = max(int(np.searchsorted(pos, e_k, side="left")), r1)
        out.append((lo, hi, Window(w0, e_k, stat_end, 0, r0 - lo, r1 - lo)))
    return out


def run(engines, hb, n_windows=None):
    """The whole contig through len(engines) windows (engine k takes window k; fewer windows when the contig has
    too few distinct read starts).  Returns (qualities, read flags, statistics) of the contig."""
    wins = plan(hb, n_windows or len(engines))
    q = np.empty(max(hb.n_bases, 1), np.uint8)[:hb.n_bases]
    f = np.zeros(hb.n_reads, np.uint8)
    stats = np.zeros(ST_N, np.int64)
    if not wins:
        return q, f, stats
    assert len(wins) <= len(engines)
    slices = []
    for eng, (lo, hi, w) in zip(engines, wins):
        sb = _slice(hb, lo, hi)
        slices.append(sb)
        eng.upload(sb)
        eng.set_window(w)
    used = engines[:len(wins)]
    for eng in used:                       # independent: one device each
        eng.run_columns()
    carry = Carry.start()
    for eng in used:                       # chained, a few small kernels per window
        nxt = Carry.start()
        eng.run_depth(carry, nxt)
        carry.total_col, carry.total_depth = nxt.total_col, nxt.total_depth
    for eng in used:
        eng.run_triggers()
    for eng in used:
        nxt = Carry.start()
        eng.run_fold(carry, nxt)
        carry.min_pos, carry.max_pos, carry.min_pos2, carry.max_pos2 = nxt.min_pos, nxt.max_pos, nxt.min_pos2, nxt.max_pos2
    for eng in used:
        eng.run_finish()
    for eng, sb, (lo, hi, w) in zip(used, slices, wins):
        wq, wf, ws = eng.download()
        r0, r1 = int(w.read_first), int(w.read_end)
        q[int(hb.seq_off[lo + r0]):int(hb.seq_off[lo + r1])] = wq[int(sb.seq_off[r0]):int(sb.seq_off[r1])]
        f[lo + r0:lo + r1] = wf[r0:r1]
        stats += ws
    return q, f, stats


def run_rank(engine, hb, dist, n_windows=None):
    """The same protocol with one process per window (torch.distributed: rank k takes window k; ranks beyond the
    number of windows idle).  The carried state travels rank to rank as two point-to-point messages (2 and 4
    integers); there is no collective on the data path.  Returns (lo, hi, Window, qualities, flags, statistics) of
    this rank's slice -- only the owned reads [read_first, read_end) of it are final -- or None for an idle rank."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    wins = plan(hb, n_windows or world)
    if rank >= len(wins):
        return None
    lo, hi, w = wins[rank]
    sb = _slice(hb, lo, hi)
    engine.upload(sb)
    engine.set_window(w)
    engine.run_columns()
    carry, nxt = Carry.start(), Carry.start()
    last = len(wins) - 1
    if rank > 0:
        t = torch.zeros(2, dtype=torch.int64)
        dist.recv(t, src=rank - 1)
        carry.total_col, carry.total_depth = int(t[0]), int(t[1])
    engine.run_depth(carry, nxt)
    if rank < last:
        dist.send(torch.tensor([nxt.total_col, nxt.total_depth], dtype=torch.int64), dst=rank + 1)
    engine.run_triggers()
    if rank > 0:
        t = torch.zeros(4, dtype=torch.int64)
        dist.recv(t, src=rank - 1)
        carry.min_pos, carry.max_pos, carry.min_pos2, carry.max_pos2 = (int(x) for x in t)
    engine.run_fold(carry, nxt)
    if rank < last:
        dist.send(torch.tensor([nxt.min_pos, nxt.max_pos, nxt.min_pos2, nxt.max_pos2], dtype=torch.int64), dst=rank + 1)
    engine.run_finish()
    q, f, s = engine.download()
    return lo, hi, w, q, f, s
