"""README.methods' STR sketch (/root/reference/README.methods:156-178) run through the path.

The sketch draws the 25-base sequence AGCTCCGTATATAGGACTGAGGTAG with a variant X on the second base of the TATATA
repeat and the bases whose qualities are kept at levels 1, 3 and 9.  It is the author's illustration, not program
output -- the section ends "TODO: check this with real examples!" -- and its arithmetic ("STR len 4 => keep 2*4+1") is
the INDEL rule of level 1 applied to a column it calls a SNP.  So the drawing is not a golden vector; what is pinned
here is what the code it illustrates does with that very sequence:

  SNP at X (index 8), call not confident (5 of 30 reads, Q20..Q29):
    level 9  sSTR 0.0,0   -> the column itself: [8, 8]          == the sketch's level-9 row, cell for cell
    level 3  sSTR 1.0,0   -> the STR under it:  [7, 12]         (sketch: [6, 13]; one base wider either side)
    level 1  sSTR 1.0,5   -> every STR whose span comes within 5 of X: CC (4..5) .. GG (13..14), so
                             [8 - (8-4) - 5, 8 + (14-8) + 5] = [-1, 19]   (sketch: [5, 17])
    (mask_LC_regions' min_pos / max_pos over the STRs near the column, snp_score.c:1232-1275, then
     pos -/+ (pos - min_pos) * mul -/+ add, snp_score.c:1745-1755)
  2-base deletion of a TA unit after X: iSTR 2.0,1 / 1.1,2 / 1.0,2 -- zones recorded from the oracle (they include the
    back-fill of the deletion's own columns and the reads' STR-spanning test) and checked to be the same for every
    flank draw: the rule depends on the 25 bases only.

Every zone is read off the OUTPUT qualities (a base is "kept" when every read covering it still has its original
quality; reads kept whole by another rule aside), the oracle first, then the emulated kernel sources and the HIP engine through the C ABI."""
import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import HostBatch
import oracle_lib as O
import manpage_example as M

S = "AGCTCCGTATATAGGACTGAGGTAG"
X = 8
FL = 120
ALT = [3, 9, 14, 20, 25, 7, 11, 17]           # which of the 30 reads carry the variant
SNP_ZONES = {1: (-1, 19), 3: (7, 12), 9: (8, 8)}
DEL_ZONES = {1: (3, 20), 3: (4, 16), 9: (5, 16)}
SKETCH = {1: (5, 17), 3: (6, 13), 9: (8, 8)}  # the README's drawing, for the record


def build(seed, kind, nalt, alt=ALT):
    rng = np.random.default_rng(seed)
    left = M._flank(rng, FL, avoid_first=S[:4][::-1])[::-1]
    right = M._flank(rng, FL, avoid_first=S[-4:])
    full = left + S + right
    starts = sorted(int(x) for x in rng.integers(0, FL - 20, 30))
    reads, maps = [], []
    for i, s in enumerate(starts):
        e = int(rng.integers(FL + len(S) + 20, len(full)))
        seq = list(full[s:e])
        cig = f"{e - s}M"
        qmap = {k: FL + k - s for k in range(-20, len(S) + 20)}        # sketch index -> query index
        if i in alt[:nalt]:
            if kind == "snp":
                seq[FL + X - s] = "C"
            else:
                k = FL + X - s + 1
                del seq[k:k + 2]
                cig = f"{k}M2D{e - s - k - 2}M"
                qmap = {j: (q if j <= X else q - 2) for j, q in qmap.items() if j not in (X + 1, X + 2)}
        reads.append(dict(pos=1000 + s, cigar=cig, seq="".join(seq), qual=[int(x) for x in rng.integers(20, 30, len(seq))],
                          mapq=60))
        maps.append(qmap)
    return HostBatch.from_reads(reads), maps


def zone(hb, maps, q):
    kept, tot = {}, {}
    for r, qmap in enumerate(maps):
        o = int(hb.seq_off[r])
        n = int(hb.seq_off[r + 1]) - o
        if np.array_equal(q[o:o + n], hb.quals[o:o + n]):
            continue                  # a read kept as a whole (the over-depth rule on a pile of equal starts, seed 3)
        for j, qi in qmap.items():
            if 0 <= qi < n:
                tot[j] = tot.get(j, 0) + 1
                kept[j] = kept.get(j, 0) + int(q[o + qi] == hb.quals[o + qi])
    full = sorted(j for j in tot if kept[j] == tot[j])
    part = [j for j in tot if 0 < kept[j] < tot[j]]
    assert not part, part                                              # a column is kept for every read or for none
    if not full:
        return None
    assert full == list(range(full[0], full[-1] + 1))
    return full[0], full[-1]


def check(run, seeds=range(4)):
    for level in (1, 3, 9):
        p = ca.preset_params(level, pblock=0)
        for seed in seeds:
            hb, maps = build(seed, "snp", 5)
            assert zone(hb, maps, run(p, hb)) == SNP_ZONES[level], (level, seed)
            hb, maps = build(seed, "del", 3)
            assert zone(hb, maps, run(p, hb)) == DEL_ZONES[level], (level, seed)


def test_sketch_rows_as_drawn():
    """the drawing itself, so the constants above can be checked against the README by eye"""
    rows = {1: "     qqqqqqqqqqqqq", 3: "      qqqqqqqq", 9: "        q"}
    for lv, row in rows.items():
        assert (row.index("q"), len(row) - 1) == SKETCH[lv]
    assert "       -X----".index("X") == X and S[7:13] == "TATATA"
    assert SNP_ZONES[9] == SKETCH[9]


def test_oracle_on_the_sketch():
    check(lambda p, hb: O.contig(p, hb)[0])


def test_a_confident_snp_keeps_nothing():
    """15 of 30 reads: a clean heterozygous call -- no quality within 20 bases of the sketch survives at any level (the
    sketch's X has to be a call crumble is unsure of)"""
    for level in (1, 3, 9):
        hb, maps = build(0, "snp", 15, alt=list(range(1, 30, 2)))
        q = O.contig(ca.preset_params(level, pblock=0), hb)[0]
        assert zone(hb, maps, q) is None


def test_emulated_kernels_on_the_sketch():
    import emu_lib

    def run(p, hb):
        e = emu_lib.engine(p)
        try:
            return e.contig(hb)[0]
        finally:
            e.close()
    check(run, seeds=range(2))


@pytest.mark.gpu
def test_hip_engine_on_the_sketch():
    def run(p, hb):
        e = ca.Engine(p, 0)
        try:
            return e.contig(hb)[0]
        finally:
            e.close()
    check(run)
