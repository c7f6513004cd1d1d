"""The reference's own published before / after example (crumble.1 EXAMPLES, /root/reference/crumble.1:516-600) as a
golden fixture: the first piece of REAL crumble output this code is compared with beyond find_STR.  See
tests/manpage_example.py for what is asserted and which cells are excluded; tests/golden/make_manpage_example.py made
the JSON from the man page.

CPU tier: the oracle (and the emulated kernel sources); GPU tier: the HIP engine through the C ABI."""
import numpy as np
import pytest

import crumble_amd as ca
import oracle_lib as O
import manpage_example as M

SEEDS = range(8)            # the flanks the man page does not show: eight different draws, every one must agree


def params(pblock):
    # crumble -9p8 -l5 -u40 -i1.0,2 (all of them -9's own values; spelled out because the page spells them out)
    return ca.preset_params(9, pblock=pblock, qlow=5, qhigh=40, iSTR_mul=1.0, iSTR_add=2)


def check_p0(run, seed):
    """-p0: kept zone exactly [1488393, 1488421] for every read, original qualities inside, Q40 / the one Q5 outside"""
    hb, cmap, d = M.build(seed)
    q = run(params(0), hb)
    got = M.cells(hb, cmap, q)
    for i, (g, w) in enumerate(zip(got, d["expected_p0"])):
        assert g == w, (seed, "read", i, g, w)
    # the page's own description: "Qualities outside this region are replaced with Q40"
    z0, z1 = d["zone"]
    n_out = 0
    for r, g in zip(d["reads"], got):
        for k, ch in enumerate(g):
            if not z0 <= r["first"] + k <= z1:
                assert ch in "I&"
                n_out += 1
    assert n_out == sum(len(x["cells"]) for x in d["reads"]) - sum(
        1 for x in d["reads"] for k in range(len(x["cells"])) if z0 <= x["first"] + k <= z1)
    assert sum(g.count("&") for g in got) == 1
    return hb, cmap, d, q


def check_pblock(pblock_fn, hb, cmap, d, q0):
    """the page's `after` block inside the zone = P-block level 8 of each read's kept run on its own"""
    segs = M.zone_segments(hb, cmap, d)
    pieces = [q0[a:b] for s in segs if s for a, b in [s]]
    so = np.concatenate([[0], np.cumsum([len(p) for p in pieces])]).astype(np.int64)
    out = pblock_fn(so, np.concatenate(pieces))
    q = q0.copy()
    k = 0
    for s in segs:
        if s:
            q[s[0]:s[1]] = out[so[k]:so[k + 1]]
            k += 1
    got = M.cells(hb, cmap, q)
    for i, (g, w) in enumerate(zip(got, d["expected_after"])):
        assert g == w, ("read", i, g, w)


def _oracle_run(p, hb):
    return O.contig(p, hb)[0]


def _oracle_pblock(so, quals):
    p = params(8)
    return np.concatenate([O.pblock(quals[so[i]:so[i + 1]], p.pblock, p.qcap) for i in range(len(so) - 1)])


def test_fixture_is_the_man_page_example():
    d = M.load()
    assert d["first"] == 1488390 and d["last"] == 1488424 and d["zone"] == [1488393, 1488421]
    assert len(d["before"]) == len(d["after"]) == 35 and len(d["reads"]) == 14
    assert d["before"][8].split()[4] == "c$CCc-1n*C-1NC+1AaCC-1NCC-1N"            # the deletion / insertion column
    assert d["after"][0].split()[5] == "II&IIIIIIII"
    assert sum(len(r["cells"]) for r in d["reads"]) == 355


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_reproduces_the_man_page(seed):
    hb, cmap, d, q = check_p0(_oracle_run, seed)
    check_pblock(_oracle_pblock, hb, cmap, d, q)


@pytest.mark.parametrize("kcol", ["v1", "v2"])
def test_emulated_kernels_reproduce_the_man_page(kcol):
    import emu_lib
    def run(p, hb):
        e = emu_lib.engine(p, kcol=kcol)
        try:
            return e.contig(hb)[0]
        finally:
            e.close()
    hb, cmap, d, q = check_p0(run, 0)
    e = emu_lib.engine(params(8))
    try:
        check_pblock(e.pblock, hb, cmap, d, q)
    finally:
        e.close()


def test_todays_p8_differs_from_the_page_only_where_runs_cross_the_zone_edge():
    """Why -p8 itself is not compared cell for cell (see manpage_example.py): with the keep flag stripped before
    pblock (snp_score.c:1093-1099) a run continues into the Q40s next to the zone.  The cells that differ from the page
    must all belong to runs that touch a zone edge; runs strictly inside the zone equal the page."""
    hb, cmap, d = M.build(0)
    q = O.contig(params(8), hb)[0]
    got = M.cells(hb, cmap, q)
    z0, z1 = d["zone"]
    ndiff = 0
    for r, g, w in zip(d["reads"], got, d["expected_after"]):
        diff = [k for k in range(len(g)) if g[k] != w[k]]
        ndiff += len(diff)
        for k in diff:
            # walk the equal-valued run of today's output around the cell: it must reach past a zone edge or the block's edge
            a = k
            while a > 0 and g[a - 1] == g[k]:
                a -= 1
            b = k
            while b + 1 < len(g) and g[b + 1] == g[k]:
                b += 1
            assert r["first"] + a <= z0 or r["first"] + b >= z1 or a == 0 or b == len(g) - 1, (r["first"], k, g, w)
    assert ndiff > 0                                                               # (read 1: 40 40 40 | 30 27 ... is one run today)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_hip_engine_reproduces_the_man_page(seed):
    def run(p, hb):
        e = ca.Engine(p, 0)
        try:
            return e.contig(hb)[0]
        finally:
            e.close()
    hb, cmap, d, q = check_p0(run, seed)
    e = ca.Engine(params(8), 0)
    try:
        check_pblock(e.pblock, hb, cmap, d, q)
        for opt in ("reads", "segments", "words"):                                        # both P-block decompositions
            e.set_option("pblock", opt)
            check_pblock(e.pblock, hb, cmap, d, q)
    finally:
        e.close()
