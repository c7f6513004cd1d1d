#!/usr/bin/env python3
"""GPU box: long randomized parity run (tests/fuzz_util.py cases + tiny-read batches), engine == oracle.
Usage: python tools/gpu_fuzz.py <first_seed> <n_seeds>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import crumble_amd as ca
from fuzz_util import random_case
from parity_util import assert_parity
from test_fuzz_parity import tiny_reads_batch
s0, n = int(sys.argv[1]), int(sys.argv[2])
bad, t0 = 0, time.time()
for seed in range(s0, s0 + n):
    p, hb, desc = random_case(seed, max_bases=300_000)
    kcol = ("v1", "v1perm", "v2", "v2", "auto")[seed % 5]       # every column kernel, whatever the batch
    opts = {"kcol": kcol, "pblock": ("auto", "reads", "segments", "words")[(seed // 5) % 4],   # ... and every P-block decomposition
            "run": ("fast", "fast", "careful")[seed % 3], "caps": (0, 0, 0, 0, 0, 0, 4)[seed % 7],   # ... and the run's fast way, the careful one, an aborted attempt
            "replay": ("auto", "auto", "auto", "flat")[(seed // 3) % 4]}                          # ... and the byte-wise replay
    if seed % 2:
        hb.pack4(ca.lib())                                      # the bases as BAM packs them (crumble_batch.bases4)
    if seed % 4 >= 2:
        hb.lens()                                               # BAM's per-read lengths instead of the offset tables
    if seed % 3 == 1:                                           # qualities of a binned instrument, sent as 2- / 4-bit codes (k_p0_quals)
        import numpy as np
        vals = np.array([2, 11, 25, 37, 6, 15, 22, 27, 33, 40, 41, 8, 30, 19, 35, 39], np.uint8)[:(4, 8, 16, 3)[(seed // 3) % 4]]
        hb.quals = vals[hb.quals % len(vals)]
        hb.packq(ca.lib())
    try:
        assert_parity(lambda q: ca.Engine(q, 0, options=opts), p, hb)
        if seed % 10 == 0:
            assert_parity(lambda q: ca.Engine(q, 0, options=opts), p, tiny_reads_batch(seed, n=2000), check_columns=False)
        if seed % 3 == 0 and hb.n_reads >= 8 and hb.tid >= 0:          # the same contig cut into windows
            import numpy as np
            import oracle_lib as O
            from crumble_amd import windows
            nw = 2 + seed % 4
            engs = [ca.Engine(p, 0) for _ in range(nw)]
            try:
                got = windows.run(engs, hb, nw)
            finally:
                for e_ in engs:
                    e_.close()
            want = O.contig(p, hb)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), "windows differ from the whole contig"
    except Exception as e:
        bad += 1
        print("FAIL", desc, str(e)[:300], flush=True)
    if (seed - s0) % 200 == 199:
        print(f"{seed - s0 + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n} cases, {bad} failures, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
