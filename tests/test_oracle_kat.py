"""Pin the CPU oracle before trusting it.

Sources of truth used here:
  * the REAL reference find_STR: /root/reference/str_finder.c compiled
    unmodified into oracle/_ref/libstrfinder_ref.so (oracle/Makefile);
  * the known answers the survey obtained by running the compiled reference
     for calculate_consensus_pileup (snp_score.c:533),
    pblock (snp:803), ref2query_pos (snp:1156), bam_qpos2rpos (snp:1205),
    mask_LC_regions (snp:1230) and find_STR;
  * the documented examples str_finder.h:31-35.
"""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
from crumble_amd import HostBatch, preset_params

A, Cc, G, T, N = 1, 2, 4, 8, 15       # nt16 codes
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def col(params, flags, spec, q=30, mq=60):
    """spec: list of (count, base16 or '*')"""
    b, d = [], []
    for cnt, base in spec:
        for _ in range(cnt):
            if base == "*":
                b.append(A); d.append(1)
            else:
                b.append(base); d.append(0)
    n = len(b)
    return O.consensus(params, flags, b, [q] * n, [mq] * n, d)


# (spec, flags, q, mapq) -> call, het_call, het_phred, phred, discrep
D2 = [
    ([(30, A)], 20, 30, 60, (0, 1, -151, 146, 0.0)),
    ([(30, A)], 4, 30, 60, (0, 1, -152, 147, 0.0)),
    ([(15, Cc), (15, A)], 20, 30, 60, (0, 1, 142, 0, 0.0)),
    ([(2, Cc), (28, A)], 20, 30, 60, (0, 1, -111, 112, 0.357806)),
    ([(5, Cc), (25, A)], 20, 30, 60, (0, 1, -52, 53, 0.894515)),
    ([(10, "*"), (20, A)], 20, 30, 60, (0, 4, 46, 0, 0.0)),
    ([(1, N), (29, A)], 20, 30, 60, (0, 1, -148, 143, 0.0)),
    ([(30, N)], 20, 30, 60, (5, 0, 0, 0, 0.0)),
    ([(30, A)], 20, 30, 0, (0, 1, -68, 0, 0.0)),
    ([(300, A)], 20, 40, 60, (0, 1, -959, 255, 0.0)),
    ([(150, G), (150, A)], 20, 40, 60, (0, 2, 2168, 0, 0.0)),
]


@pytest.mark.parametrize("spec,flags,q,mq,want", D2)
def test_consensus_known_answers(product, spec, flags, q, mq, want):
    p = preset_params(9)
    r = col(p, flags, spec, q, mq)
    got = (r["call"], r["het_call"], r["het_phred"], r["phred"])
    assert got == want[:4]
    assert abs(r["discrep"] - want[4]) < 1e-6


def test_consensus_depth_series(product):
    """nA at Q30, mode B: het_phred / phred by depth."""
    p = preset_params(9)
    het = {1: -65, 3: -71, 5: -77, 8: -85, 10: -92, 12: -98, 15: -106}
    ph = {1: 14, 3: 53, 5: 72, 8: 80, 10: 87, 12: 93, 15: 101}
    for d in het:
        r = col(p, 20, [(d, A)])
        assert (r["call"], r["het_call"], r["het_phred"], r["phred"]) == (0, 1, het[d], ph[d]), d
    # clean hom column phred by depth for several qualities (mode B, mapq 60)
    series = {
        30: {1: 14, 2: 33, 3: 53, 4: 67, 5: 72, 6: 75, 7: 77, 8: 80, 9: 83, 10: 87, 15: 101,
             20: 116, 25: 131, 30: 146, 35: 161, 40: 176},
        37: {1: 17, 2: 40, 3: 61, 4: 69, 5: 72, 6: 75, 7: 78, 8: 81, 10: 87, 20: 116, 30: 146,
             40: 176},
        40: {1: 19, 2: 43, 3: 64, 4: 69, 5: 72, 6: 75, 30: 147, 40: 177},
        93: {1: 31, 2: 61, 3: 66, 4: 69, 5: 72, 6: 75, 30: 147, 40: 177},
    }
    p.qcap = 100
    for q, tab in series.items():
        for d, want in tab.items():
            r = col(p, 20, [(d, A)], q=q)
            assert r["phred"] == want, (q, d, r)


def test_pblock_known_answers(product):
    q = [30, 31, 33, 36, 40, 12, 12, 13, 30, 30, 45, 46, 47, 2, 2, 41]
    assert O.pblock(q, 8, 60).tolist() == [35, 35, 35, 35, 35, 12, 12, 12, 38, 38, 38, 38, 47, 2, 2, 41]
    assert O.pblock(q, 2, 60).tolist() == [31, 31, 31, 38, 38, 12, 12, 12, 30, 30, 46, 46, 46, 2, 2, 41]
    # last run is not capped (Q12)
    assert O.pblock([70, 72, 74, 76, 10, 10, 66, 68], 8, 60).tolist() == \
        [60, 60, 60, 60, 10, 10, 67, 67]
    assert O.pblock([], 8, 60).tolist() == []
    assert O.pblock([7], 8, 60).tolist() == [7]


def test_pblock_preserve_breaks_runs(product):
    pq = np.zeros(256, np.uint8); pq[93] = 1
    # 93 is skipped and splits the run; a trailing preserve run touches only qual[len]
    assert O.pblock([30, 32, 93, 93, 30, 34], 8, 60, pq).tolist() == [31, 31, 93, 93, 32, 32]
    assert O.pblock([30, 32, 93], 8, 60, pq).tolist() == [31, 31, 93]
    assert O.pblock([93, 30, 32], 8, 60, pq).tolist() == [93, 31, 31]


READ = dict(pos=100, cigar="5S10M2I10M3D10M4S", seq="A" * 41, qual=30)


def test_ref2query_pos_known_answers(product):
    hb = HostBatch.from_reads([READ])
    want = {90: 0, 100: 0, 101: 6, 109: 14, 110: 15, 111: 18, 119: 26, 120: 27, 121: 27,
            122: 27, 123: 27, 132: 36, 133: 37, 200: 41}
    for pos, q in want.items():
        assert O.ref2query_pos(hb, 0, pos) == q, pos


def test_qpos2rpos_known_answers(product):
    hb = HostBatch.from_reads([READ])
    want = {0: 100, 4: 100, 5: 100, 6: 101, 14: 109, 15: 110, 16: 110, 17: 110, 18: 111,
            26: 119, 27: 120, 28: 122, 36: 132, 37: 133, 40: 133, 41: 133}
    for q, r in want.items():
        assert O.qpos2rpos(hb, 0, q) == r, q


def test_find_str_known_answers(product):
    assert O.find_str("AGGGGAGGAGAAGAC") == [(1, 4), (3, 9), (7, 10), (8, 13)]
    assert O.find_str("AGCTCCGTATATAGGACTGAGGTAG") == [(4, 5), (7, 12), (13, 14), (20, 21)]
    assert O.find_str("GATCGGACGAGAGAGAGAGAGTCGGAC") == [(4, 5), (8, 20), (23, 24)]
    assert O.find_str("") == []
    assert O.find_str("A") == []


def test_mask_lc_q7_known_answers(product):
    """SURVEY.md App. D.2 'Q7 confirmed': the nibble one past the read end."""
    p = preset_params(9, iSTR_add=2)
    INT_MAX = 2**31 - 1
    for q0, want in ((20, (1013, 1019)), (50, (1013, 1019)), (70, (1013, 1019)),
                     (40, (1013, 1020))):
        qual = [q0] + [30] * 19
        hb = HostBatch.from_reads([dict(pos=1000, cigar="20M", seq="GATTGCTAGGCTTACACACA",
                                        qual=qual)])
        assert O.mask_lc(p, hb, 0, 1, 16, INT_MAX, 0) == want, q0
    seq19 = "GATTGCTAGGCTTGCAAAA"
    hb = HostBatch.from_reads([dict(pos=1000, cigar="19M", seq=seq19, qual=30)])
    assert O.mask_lc(p, hb, 0, 1, 17, INT_MAX, 0) == (1015, 1019)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_find_str_matches_real_reference(product):
    """Restated find_STR/add_rep == reference str_finder.c on random + repetitive strings."""
    rng = random.Random(12345)
    alph = "ACGT"
    n_checked = 0
    for it in range(4000):
        L = rng.choice([1, 2, 3, 7, 14, 15, 16, 17, 30, 60, 120, 301, 501])
        mode = it % 4
        if mode == 0:
            s = "".join(rng.choice(alph) for _ in range(L))
        elif mode == 1:      # low-complexity: 2-letter alphabet
            s = "".join(rng.choice("AC") for _ in range(L))
        elif mode == 2:      # planted tandem repeats
            s = ""
            while len(s) < L:
                if rng.random() < 0.5:
                    unit = "".join(rng.choice(alph) for _ in range(rng.randint(1, 9)))
                    s += unit * rng.randint(2, 12)
                else:
                    s += "".join(rng.choice(alph) for _ in range(rng.randint(1, 10)))
            s = s[:L]
        else:                # IUPAC / N / '=' all behave like A (Q8)
            s = "".join(rng.choice("ACGTNN=MRK") for _ in range(L))
        assert O.find_str(s) == O.ref_find_str(s), s
        n_checked += 1
    assert n_checked == 4000


def test_find_str_golden(product):
    """Committed vectors generated from the real reference (tests/golden/make_str_golden.py)."""
    with open(os.path.join(GOLD, "find_str.json")) as f:
        gold = json.load(f)
    assert len(gold) >= 200
    for s, want in gold:
        assert O.find_str(s) == [tuple(x) for x in want], s


def test_adjust_lut_matches_formula(product):
    """(qual, mapq) -> adjusted quality: spot values derived from App. D.2 behaviour."""
    lut = O.adjust_lut()
    assert lut.shape == (256, 256)
    # mapq 0 => m = 1-10^(-0.005): adjusted quality collapses to ~1 (30xA mapq0 -> phred 0)
    assert lut[0, 30] <= 2
    # monotone in qual for high mapq, and never above the input quality + 1
    row = lut[60, :94].astype(int)
    assert (np.diff(row) >= 0).all()
    assert lut[255, 30] == lut[10, 30] == lut[11, 30]      # mqual_pow[255] = mqual_pow[10]; i/2 integer
