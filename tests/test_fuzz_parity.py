"""Randomised parity: random option sets x random synthetic alignments, engine == oracle bit for bit
(quality bytes, per-read flags, statistics, column records).  CPU tier: the emulated kernels on a
few seeds; GPU tier: the real build on many."""
import pytest

import crumble_amd as ca
from fuzz_util import random_case
from parity_util import assert_parity


@pytest.mark.parametrize("seed", range(100, 112))
def test_fuzz_emulated(product, seed):
    import emu_lib
    p, hb, desc = random_case(seed, max_bases=120_000)
    print(desc)
    assert_parity(emu_lib.engine, p, hb)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(6))
def test_fuzz_gpu(product, block):
    for seed in range(1000 + 25 * block, 1000 + 25 * (block + 1)):
        p, hb, desc = random_case(seed, max_bases=400_000)
        try:
            assert_parity(lambda q: ca.Engine(q, 0), p, hb)
        except AssertionError as e:
            raise AssertionError(f"{desc}: {e}") from e
