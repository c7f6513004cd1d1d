"""htslib bam_plp_auto semantics on hand-derived CIGAR cases.

htslib is not vendored by the reference and is absent from the image, so this
part of the oracle is *unpinned* (DESIGN.md): the expectations below are derived
by hand from htslib's resolve_cigar2 rules.  The same reads then go through the
engine (emulated kernels here, real HIP in tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import crumble_amd as ca
from crumble_amd import HostBatch
import oracle_lib as O
import emu_lib
from parity_util import assert_parity

SEQ = "ACGTACGTACGTACGTACGTACGTACGTACGT"


def plp(cigar, lq, pos=100):
    hb = HostBatch.from_reads([dict(pos=pos, cigar=cigar, seq=SEQ[:lq], qual=30)])
    return O.pileup(hb)


def cols(ent):
    return [e["pos"] for e in ent]


def test_plain_match():
    e = plp("10M", 10)
    assert cols(e) == list(range(100, 110))
    assert [x["qpos"] for x in e] == list(range(10))
    assert e[0]["is_head"] and not e[0]["is_tail"] and e[-1]["is_tail"] and not e[-1]["is_head"]
    assert all(x["indel"] == 0 and not x["is_del"] and not x["is_refskip"] for x in e)


def test_soft_clips_and_hard_clip_and_leading_insertion():
    e = plp("5S10M5S", 20)
    assert [x["qpos"] for x in e] == list(range(5, 15)) and cols(e) == list(range(100, 110))
    e = plp("3H5M", 5)
    assert [x["qpos"] for x in e] == [0, 1, 2, 3, 4]
    e = plp("2I8M", 10)
    assert [x["qpos"] for x in e] == list(range(2, 10))


def test_insertion():
    e = plp("5M2I5M", 12)
    assert cols(e) == list(range(100, 110))
    assert [x["qpos"] for x in e] == [0, 1, 2, 3, 4, 7, 8, 9, 10, 11]
    assert [x["indel"] for x in e] == [0, 0, 0, 0, 2, 0, 0, 0, 0, 0]
    e = plp("8M2I", 10)
    assert e[-1]["indel"] == 2 and e[-1]["is_tail"] and cols(e) == list(range(100, 108))
    e = plp("5M1P2I5M", 12)
    assert [x["indel"] for x in e][4] == 2


def test_deletions():
    e = plp("5M2D5M", 10)
    assert cols(e) == list(range(100, 112))
    assert [x["indel"] for x in e] == [0, 0, 0, 0, -2, 0, 0, 0, 0, 0, 0, 0]
    assert [x["is_del"] for x in e] == [0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0]
    assert [x["qpos"] for x in e] == [0, 1, 2, 3, 4, 5, 5, 5, 6, 7, 8, 9]     # deleted columns report the NEXT base
    e = plp("5M1D2D5M", 10)
    assert [x["indel"] for x in e][:8] == [0, 0, 0, 0, -3, 0, 0, 0]
    assert [x["is_del"] for x in e][:9] == [0, 0, 0, 0, 0, 1, 1, 1, 0]
    e = plp("5M2D2I5M", 12)
    assert [x["indel"] for x in e][:8] == [0, 0, 0, 0, -2, 0, 2, 0]
    assert [x["qpos"] for x in e][:9] == [0, 1, 2, 3, 4, 5, 5, 7, 8]


def test_ref_skip():
    e = plp("5M100N5M", 10)
    assert cols(e) == list(range(100, 210))
    assert all(x["is_refskip"] and x["is_del"] and x["qpos"] == 5 for x in e[5:105])
    assert e[4]["indel"] == 0 and e[105]["qpos"] == 5 and not e[105]["is_del"]


def test_no_ref_op_read_is_not_in_pileup():
    hb = HostBatch.from_reads([dict(pos=100, cigar="5S20I", seq=SEQ[:25], qual=30),
                               dict(pos=100, cigar="10M", seq=SEQ[:10], qual=30, flag=1024)])
    assert O.pileup(hb) == []


def test_column_order_is_file_order():
    reads = [dict(pos=100, cigar="10M", seq=SEQ[:10]), dict(pos=100, cigar="4M", seq=SEQ[:4]),
             dict(pos=103, cigar="10M", seq=SEQ[:10]), dict(pos=120, cigar="5M", seq=SEQ[:5])]
    e = O.pileup(HostBatch.from_reads(reads))
    at103 = [x["read"] for x in e if x["pos"] == 103]
    assert at103 == [0, 1, 2]
    assert 110 in cols(e) and 113 not in cols(e) and 119 not in cols(e) and 120 in cols(e)


CASES = ["10M", "5S10M5S", "5M2I5M", "5M2D5M", "5M1D2D5M", "5M2D2I5M", "5M1P2I5M", "5M100N5M",
         "3H5M", "2I8M", "8M2I", "4M3I2D6M", "6M1D1I6M", "1M", "5S20I", "12M3S", "3M200N3M1I4M"]


@pytest.mark.parametrize("level", [1, 5, 9])
def test_engine_on_cigar_zoo_emulated(product, level):
    from parity_util import cigar_zoo
    hb = cigar_zoo(CASES)
    assert_parity(emu_lib.engine, ca.preset_params(level), hb)
    assert_parity(emu_lib.engine, ca.preset_params(level, keep="30-33", softclip=1), hb)
