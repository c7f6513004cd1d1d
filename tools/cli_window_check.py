"""Runs ON THE GPU BOX: crumble-hip -j 1 against -j 4 -w and -j 7 -w (all engines on device 0) on a 3.1 Mb short-read
contig plus a 0.8 Mb HiFi contig at -9: output qualities, -b lines and statistics must be identical.
Usage: python tools/cli_window_check.py"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import synth
from crumble_amd import bamio
tmp = tempfile.mkdtemp()
refs = [("chrA", 3_200_000), ("chrB", 900_000)]
bs = [synth.generate(synth.config("chr20", scale=1.0, contig_len=3_100_000, seed=5, tid=0)),
      synth.generate(synth.config("hifi", scale=1.0, contig_len=800_000, seed=6, tid=1))]
bamio.write_batches(os.path.join(tmp, "in.bam"), refs, bs)
exe = "crumble_amd/crumble-hip"
outs = {}
for tag, args, devs in (("one", ["-j", "1"], "0"), ("win4", ["-j", "4", "-w"], "0,0,0,0"), ("win7", ["-j", "7", "-w"], "0,0,0,0,0,0,0")):
    r = subprocess.run([exe, "-v", "-9", "-b", os.path.join(tmp, tag + ".bed")] + args + [os.path.join(tmp, "in.bam"), os.path.join(tmp, tag + ".bam")],
                       stderr=subprocess.PIPE, text=True, env=dict(os.environ, CRUMBLE_HIP_DEVICES=devs))
    assert r.returncode == 0, r.stderr
    st = [l for l in r.stderr.splitlines() if l.split() and l.split()[0] in ("columns", "bases", "bases_changed", "triggers", "reads", "hom_B", "indel", "over_depth")]
    rd = bamio.Reader(os.path.join(tmp, tag + ".bam"))
    q = np.concatenate([g.quals for g in rd])
    outs[tag] = (q, open(os.path.join(tmp, tag + ".bed")).read(), st)
    print(tag, len(q), [l for l in r.stderr.splitlines() if "end-to-end" in l])
for tag in ("win4", "win7"):
    print(tag, "quals equal", np.array_equal(outs[tag][0], outs["one"][0]), "bed equal", outs[tag][1] == outs["one"][1], "stats equal", outs[tag][2] == outs["one"][2])
