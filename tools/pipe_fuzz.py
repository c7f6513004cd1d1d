#!/usr/bin/env python3
"""Runs ON THE GPU BOX: crumble-hip's device pipeline against its host pipeline on random files.  Every case is a BAM of 1-9
contigs drawn from tests/fuzz_util.py's generator (short / long reads, panel depth spikes, soft clips, reference skips, mapq 0,
unmapped mates) plus, sometimes, an unmapped tail and records without qualities; written at a random level and sometimes
re-cut into small BGZF members; run with a random option set (presets, -k / -K, -y pbccs, -p, keep BED) through
CRUMBLE_HIP_PIPELINE=host and =device (random read-buffer and drain sizes, 1-4 engines); the inflated outputs must be equal
byte for byte and the statistics blocks too.  Every fifth case also against the oracle.
Usage: tools/pipe_fuzz.py [first_seed] [seconds] [exe]"""
import gzip
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import crumble_amd as ca
from crumble_amd import bamio
from fuzz_util import random_case
import test_bamio as TB

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
exe = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "crumble_amd", "crumble-hip")
tmp = os.environ.get("TMPDIR", "/tmp")
src, bedf = os.path.join(tmp, "pf_in.bam"), os.path.join(tmp, "pf_keep.bed")
OPTS = [(["-5"], dict(level=5)), (["-1"], dict(level=1)), (["-9"], dict(level=9)), (["-9", "-p", "8"], dict(level=9, pblock=8)),
        (["-7", "-k", "2,11"], dict(level=7, keep="2,11")), (["-8", "-K", "37"], dict(level=8, really_keep="37")),
        (["-3", "-y", "pbccs"], dict(level=3, tech="pbccs")), (["-5", "-u", "35", "-c", "25"], dict(level=5, qhigh=35, qcutoff=25))]
t0, cases, bad, fell_back = time.time(), 0, 0, 0
seed = seed0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    nc = int(rng.integers(1, 10))
    refs, bs = [], []
    for tid in range(nc):
        _, hb, _ = random_case(seed * 16 + tid, max_bases=int(rng.choice([3000, 40000, 200000])))
        hb.bed_start = hb.bed_end = np.zeros(0, np.int32)
        hb.tid = tid
        ln = int(hb.pos.max()) + 40000 if hb.n_reads else 5000
        refs.append((f"c{tid}", ln))
        if rng.random() < 0.15:
            continue                                            # a contig without reads
        bs.append(hb)
    if rng.random() < 0.35:                                     # a long reference list (hs38DH has 3366): a refID test alone is weak then
        refs += [(f"x{t}", 5000) for t in range(int(rng.choice([300, 1000, 3500])))]
    if rng.random() < 0.5:
        bs.append(TB.three_groups()[1][2])                      # unmapped tail
    bamio.write_batches(src, refs, bs, level=int(rng.choice([0, 1, 6])))
    if rng.random() < 0.4:
        raw = gzip.decompress(open(src, "rb").read())
        open(src, "wb").write(TB.py_bgzf(raw, block=int(rng.choice([700, 5000, 30000, 65000]))))
    args, pk = OPTS[int(rng.integers(0, len(OPTS)))]
    args = list(args)
    bed = None
    if rng.random() < 0.3 and bs and bs[0].tid >= 0 and bs[0].n_reads:
        lo = int(bs[0].pos.min())
        open(bedf, "w").write("".join(f"c{bs[0].tid}\t{lo + 50 * k}\t{lo + 50 * k + 30}\n" for k in range(int(rng.integers(1, 6)))))
        args += ["-R", bedf]
        rd = bamio.Reader(src); bed = rd.load_bed(bedf); rd.close()
    out, stats = {}, {}
    for mode in ("host", "device"):
        env = dict(os.environ, CRUMBLE_HIP_PIPELINE=mode, CRUMBLE_BAM_LEVEL=str(rng.choice([0, 1, -1])))
        if mode == "device":
            env["CRUMBLE_HIP_ENGINES"] = str(rng.integers(1, 5))
            env["CRUMBLE_HIP_ASYNC_DRAIN"] = str(rng.integers(0, 2))
            if rng.random() < 0.7:
                env["CRUMBLE_HIP_DP_CHUNK"] = str(int(rng.choice([70000, 200000, 3000000])))
                env["CRUMBLE_HIP_DP_PIECE"] = str(int(rng.choice([20000, 400000, 5000000])))
        dst = os.path.join(tmp, f"pf_{mode}.bam")
        r = subprocess.run([exe, "-v", "-z", *args, src, dst], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode:
            out[mode] = ("rc", r.returncode, r.stderr[-200:])
            continue
        data = open(dst, "rb").read()
        out[mode] = gzip.decompress(data) if data[-28:] == TB.BGZF_EOF else b"no EOF member"
        stats[mode] = [l for l in r.stderr.splitlines() if len(l.split()) == 2 and l.split()[1].isdigit()]
        if mode == "device" and "host pipeline runs" in r.stderr:
            fell_back += 1
            print("seed", seed, "fell back:", [l for l in r.stderr.splitlines() if "device pipeline" in l][:1], flush=True)
        if mode == "device" and "pipeline: device" not in r.stderr and "host pipeline runs" not in r.stderr:
            out[mode] = "the device pipeline did not run"
    ok = out["host"] == out["device"] and stats.get("host") == stats.get("device") and not isinstance(out["host"], tuple)
    if ok and cases % 5 == 0 and bs:
        kw = dict(pk)
        want = TB._oracle_file(ca.preset_params(kw.pop("level"), **kw), src, bed)
        ok = np.array_equal(TB._file_quals(os.path.join(tmp, "pf_device.bam")), want)
    if not ok:
        bad += 1
        print("FAIL seed", seed, args, [type(v).__name__ if isinstance(v, bytes) else v for v in out.values()], flush=True)
    cases += 1
    seed += 1
print(f"pipe_fuzz seeds {seed0}..{seed - 1}: {cases} cases, {bad} failures, {fell_back} taken by the host pipeline instead, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
