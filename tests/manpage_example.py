"""The reference's only published input -> output pair for transcode(): the man page's mpileup of a heterozygous
deletion in a TCTC / poly-A STR before and after `crumble -9p8 -l5 -u40 -i1.0,2` (crumble.1 EXAMPLES;
tests/golden/crumble1_example.json, made by tests/golden/make_manpage_example.py).

build(seed) turns the fixture's 14 reads into a batch: the 35 printed columns exactly as printed (bases, qualities,
deletions, the insertion, read starts / ends, the printed mapqs) plus what the man page does not show -- flanks left of
1488390 and right of 1488424 (a random non-repetitive sequence every read agrees with, Q30..Q40, lengths drawn per
seed), mapq 60 for reads that start before the block, the inserted base's quality = its left neighbour's.

check(...) compares an engine's / the oracle's output with the man page, cell by cell:
  * `-9 -p0 -l5 -u40 -i1.0,2`: inside the kept zone [1488393, 1488421] every printed quality is the ORIGINAL one,
    outside it the `after` block's 'I' (Q40) -- or '&' (Q5) for the one mismatching base.  That is trigger ->
    mask_LC_regions -> iSTR growth -> back-fill / window -> qhigh / qlow (snp_score.c:1715-1919), decided by real
    crumble output.
  * P-block: inside the zone the `after` block equals pblock(level 8) (snp_score.c:803-834) of each read's kept run on
    its own.  (The crumble that made the man page blocked the kept bases while they still carried the 0x80 keep flag,
    so runs broke at the zone's edges; today's flush_bam_list strips the flag first, snp_score.c:1093-1099, and lets a
    run continue into the neighbouring Q40s -- that is why the -p8 output of the current code cannot equal the page
    cell for cell, e.g. read 1: 40 40 40 | 30 27 30 27 31 29 is one run today.)  Checked through the pblock operator
    seam on each read's zone segment.
"""
import json
import os

import numpy as np

from crumble_amd import HostBatch

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "crumble1_example.json")
FLANK = 160


def load():
    return json.load(open(FIXTURE))


def _flank(rng, n, avoid_first=None):
    """A sequence without any tandem repeat of two copies of a unit of 1..4 bases (so neither flank can extend or
    seed an STR): drawn base by base, rejecting a base that would complete such a repeat."""
    s = []
    if avoid_first:
        s = list(avoid_first)
    start = len(s)
    while len(s) - start < n:
        order = rng.permutation(4)
        for b in order:
            t = s + ["ACGT"[b]]
            ok = True
            for p in (1, 2, 3, 4):
                if len(t) >= 2 * p and t[-p:] == t[-2 * p:-p]:
                    ok = False
                    break
            if ok:
                s = t
                break
        else:
            s.pop()           # dead end (cannot happen often with 4 letters): back up one base
    return "".join(s[start:])


def build(seed=0):
    """-> (HostBatch, cellmap, fixture): cellmap[r] = query index of every printed cell of read r (the next base's
    index under a '*', as htslib reports it)."""
    d = load()
    rng = np.random.default_rng(1000 + seed)
    first, last = d["first"], d["last"]
    # the consensus of the printed columns, for context only (flanks must not repeat across the block's edges)
    win = []
    for pos in range(first, last + 1):
        cnt = {}
        for r in d["reads"]:
            k = pos - r["first"]
            if 0 <= k < len(r["cells"]) and r["cells"][k]["base"] != "*":
                b = r["cells"][k]["base"].upper()
                cnt[b] = cnt.get(b, 0) + 1
        win.append(max(cnt, key=cnt.get))
    win = "".join(win)
    left = _flank(rng, FLANK, avoid_first=win[:4][::-1])[::-1]           # read right to left away from the block
    right = _flank(rng, FLANK, avoid_first=win[-4:])
    nL = sorted((int(x) for x in rng.integers(30, FLANK - 5, len(d["reads"]))), reverse=True)   # file order = start order
    reads, cellmap = [], []
    for i, r in enumerate(d["reads"]):
        seq, qual, ops, cmap = [], [], [], []

        def op(ch, n):
            if ops and ops[-1][0] == ch:
                ops[-1][1] += n
            else:
                ops.append([ch, n])
        pos = r["first"]
        if not r["starts_inside"]:
            fl = nL[i]
            pos = first - fl
            seq += list(left[FLANK - fl:])
            qual += [int(x) for x in rng.integers(30, 41, fl)]
            op("M", fl)
        for k, c in enumerate(r["cells"]):
            if c["base"] == "*":
                op("D", 1)
                cmap.append(len(seq))                                   # the next base (htslib's qpos of a deletion)
                continue
            cmap.append(len(seq))
            seq.append(c["base"].upper())
            qual.append(ord(c["qual"]) - 33)
            op("M", 1)
            if c["indel"].startswith("+"):
                ins = c["indel"][2:]
                for ch in ins:
                    seq.append(ch.upper())
                    qual.append(ord(c["qual"]) - 33)                    # not printed by mpileup: the left neighbour's
                op("I", len(ins))
        if not r["ends_inside"]:
            fr = int(rng.integers(30, FLANK - 5))
            seq += list(right[:fr])
            qual += [int(x) for x in rng.integers(30, 41, fr)]
            op("M", fr)
        reads.append(dict(pos=pos - 1, cigar="".join(f"{n}{ch}" for ch, n in ops), seq="".join(seq), qual=qual,
                          mapq=r["mapq"] if r["mapq"] is not None else 60))
        cellmap.append(cmap)
    assert all(a["pos"] <= b["pos"] for a, b in zip(reads, reads[1:]))   # coordinate-sorted, pileup order kept
    return HostBatch.from_reads(reads), cellmap, d


def cells(hb, cellmap, q):
    """the printed qualities of every read as mpileup would print them from the byte array q"""
    out = []
    for r, cmap in enumerate(cellmap):
        o = int(hb.seq_off[r])
        out.append("".join(chr(int(q[o + x]) + 33) for x in cmap))
    return out


def zone_segments(hb, cellmap, d):
    """per read: (first byte, one past the last byte) of the qualities printed inside the kept zone, inserted base
    included"""
    z0, z1 = d["zone"]
    segs = []
    for r, (rd, cmap) in enumerate(zip(d["reads"], cellmap)):
        ks = [k for k in range(len(cmap)) if z0 <= rd["first"] + k <= z1 and rd["cells"][k]["base"] != "*"]
        if not ks:
            segs.append(None)
            continue
        o = int(hb.seq_off[r])
        segs.append((o + cmap[ks[0]], o + cmap[ks[-1]] + 1))
    return segs
