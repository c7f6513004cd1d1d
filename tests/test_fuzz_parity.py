"""Randomised parity: random option sets x random synthetic alignments, engine == oracle bit for bit
(quality bytes, per-read flags, statistics, column records).  CPU tier: the emulated kernels on a
few seeds; GPU tier: the real build on many."""
import numpy as np
import pytest

import crumble_amd as ca
from fuzz_util import random_case
from parity_util import assert_parity


@pytest.mark.parametrize("seed", range(100, 112))
def test_fuzz_emulated(product, seed):
    import emu_lib
    from parity_util import KCOL_VARIANTS
    p, hb, desc = random_case(seed, max_bases=120_000)
    print(desc)
    if seed % 2:
        hb.pack4(emu_lib.load())                 # the bases as BAM packs them (k_p0_bases<PACKED>) ...
    if seed % 4 >= 2:
        hb.lens()                                # ... BAM's per-read lengths instead of the offset tables
    if seed % 3 == 1:                            # ... qualities of a binned instrument, sent as 2- / 4-bit codes (k_p0_quals)
        import numpy as np
        vals = np.array([2, 11, 25, 37, 6, 15, 22, 27, 33, 40, 41, 8, 30, 19, 35, 39], np.uint8)[:(4, 8, 16, 3)[(seed // 3) % 4]]
        hb.quals = vals[hb.quals % len(vals)]
        hb.packq(emu_lib.load())
    # every column kernel, whatever the batch; the run's fast way, the careful one, and an aborted fast attempt
    # ... and, on every fourth seed, the byte-wise replay (k_replay_flat + k_restore / k_readfix / k_keepfix behind it)
    assert_parity(emu_lib.with_options(kcol=KCOL_VARIANTS[seed % 3], run=("fast", "careful")[(seed // 3) % 2], caps=(0, 0, 2)[seed % 3],
                                       replay=("auto", "auto", "auto", "flat")[seed % 4]), p, hb)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(6))
def test_fuzz_gpu(product, block):
    from parity_util import KCOL_VARIANTS
    for seed in range(1000 + 25 * block, 1000 + 25 * (block + 1)):
        p, hb, desc = random_case(seed, max_bases=400_000)
        if seed % 2:
            hb.pack4(ca.lib())                   # the bases as BAM packs them (k_p0_bases<PACKED>) ...
        if seed % 4 >= 2:
            hb.lens()                            # ... BAM's per-read lengths instead of the offset tables
        if seed % 3 == 1:                        # ... qualities of a binned instrument, sent as 2- / 4-bit codes (k_p0_quals)
            vals = np.array([2, 11, 25, 37, 6, 15, 22, 27, 33, 40, 41, 8, 30, 19, 35, 39], np.uint8)[:(4, 8, 16, 3)[(seed // 3) % 4]]
            hb.quals = vals[hb.quals % len(vals)]
            hb.packq(ca.lib())
        kcol = KCOL_VARIANTS[seed % 3] if seed % 4 else "auto"      # every column kernel, whatever the batch
        try:
            opts = {"kcol": kcol, "run": ("fast", "fast", "careful")[seed % 3], "caps": (0, 0, 0, 0, 5)[seed % 5],
                    "replay": ("auto", "auto", "auto", "flat")[(seed // 2) % 4]}
            assert_parity(lambda q: ca.Engine(q, 0, options=opts), p, hb)
        except AssertionError as e:
            raise AssertionError(f"{desc}: {e}") from e


def tiny_reads_batch(seed, n=600):
    """Reads of 0..40 bases (several per 16-byte chunk of the flat base array), with clips, indels, skips,
    unmapped and zero-length records in between: the chunk / segment / per-read code paths at their seams."""
    import numpy as np
    from crumble_amd import HostBatch
    rng = np.random.default_rng(seed)
    reads, pos = [], 50
    for i in range(n):
        L = int(rng.choice([0, 1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 40]))
        kind = rng.integers(0, 8)
        seq = "".join(rng.choice(list("ACGTN"), L, p=[.24, .24, .24, .24, .04]))
        qual = rng.integers(2, 60, L)
        flag, cigar = 0, f"{L}M"
        if L == 0:
            cigar = "*" if kind < 4 else "3D"
        elif kind == 0 and L > 6:
            a = int(rng.integers(1, L - 4)); cigar = f"{a}S{L - a}M"
        elif kind == 1 and L > 6:
            a = int(rng.integers(1, L - 4)); cigar = f"{L - a}M{a}S"
        elif kind == 2 and L > 8:
            a = int(rng.integers(2, L - 4)); cigar = f"{a}M2D{L - a}M"
        elif kind == 3 and L > 8:
            a = int(rng.integers(2, L - 5)); cigar = f"{a}M2I{L - a - 2}M"
        elif kind == 4 and L > 8:
            a = int(rng.integers(2, L - 4)); cigar = f"{a}M5N{L - a}M"
        elif kind == 5:
            flag, cigar = 4, "*"
        reads.append(dict(pos=pos, cigar=cigar, seq=seq, qual=qual, mapq=int(rng.choice([0, 3, 20, 60])), flag=flag))
        pos += int(rng.integers(0, 4))
    return HostBatch.from_reads(reads)


@pytest.mark.parametrize("seed", range(4))
def test_tiny_reads_emulated(product, seed):
    import emu_lib
    hb = tiny_reads_batch(seed)
    for i, p in enumerate((ca.preset_params(9), ca.preset_params(1, min_mqual=3), ca.preset_params(5, pblock=2, keep="30"),
                           ca.preset_params(9, softclip=1, binary_qual=1))):
        assert_parity(emu_lib.with_options(kcol=("v2", "v1")[(seed + i) & 1]), p, hb)


@pytest.mark.gpu
def test_tiny_reads_gpu(product):
    for seed in range(10, 30):
        hb = tiny_reads_batch(seed, n=3000)
        for p in (ca.preset_params(9), ca.preset_params(1, min_mqual=3), ca.preset_params(5, pblock=2, keep="30"),
                  ca.preset_params(9, softclip=1, binary_qual=1)):
            assert_parity(lambda q: ca.Engine(q, 0), p, hb)
