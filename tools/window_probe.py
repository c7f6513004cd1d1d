"""Runs ON THE GPU BOX: one contig cut into K windows, every window on its own engine (all on device 0), each of the
five phases timed per window.  Prints the serial total, the projected critical path with one device per window
(max over the parallel phases + sum over the chained ones) and checks the merged result against the whole-contig
run.  Usage: python tools/window_probe.py [workload] [scale] [preset] [K]"""
import json
import sys
import os
import time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import crumble_amd as ca
from crumble_amd import windows
import synth

name = sys.argv[1] if len(sys.argv) > 1 else "chr20"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
preset = int(sys.argv[3]) if len(sys.argv) > 3 else 5
K = int(sys.argv[4]) if len(sys.argv) > 4 else 8
hb = synth.generate(synth.config(name, scale=scale))
p = ca.preset_params(preset)
e0 = ca.Engine(p, 0)
e0.upload(hb); e0.run()
t0 = time.perf_counter(); e0.run(); whole_ms = (time.perf_counter() - t0) * 1e3
wq, wf, ws = e0.download()
e0.close()
wins = windows.plan(hb, K)
engs = [ca.Engine(p, 0) for _ in wins]
slices = []
for eng, (lo, hi, w) in zip(engs, wins):
    sb = windows._slice(hb, lo, hi); slices.append(sb)
    eng.upload(sb); eng.set_window(w)
ph = {k: [] for k in ("columns", "depth", "triggers", "fold", "finish")}
for rep in range(2):                                 # second round: warm
    for k in ph: ph[k] = []
    for eng, (lo, hi, w) in zip(engs, wins): eng.set_window(w)
    def timed(name, fn, eng):
        t = time.perf_counter(); fn(); eng.stats_read()      # (the read-back waits for the engine's stream)
        ph[name].append((time.perf_counter() - t) * 1e3)
    for eng in engs: timed("columns", eng.run_columns, eng)
    carry = windows.Carry.start()
    for eng in engs:
        nxt = windows.Carry.start(); timed("depth", lambda: eng.run_depth(carry, nxt), eng)
        carry.total_col, carry.total_depth = nxt.total_col, nxt.total_depth
    for eng in engs: timed("triggers", eng.run_triggers, eng)
    for eng in engs:
        nxt = windows.Carry.start(); timed("fold", lambda: eng.run_fold(carry, nxt), eng)
        carry.min_pos, carry.max_pos, carry.min_pos2, carry.max_pos2 = nxt.min_pos, nxt.max_pos, nxt.min_pos2, nxt.max_pos2
    for eng in engs: timed("finish", eng.run_finish, eng)
q = np.empty(hb.n_bases, np.uint8); f = np.zeros(hb.n_reads, np.uint8); s = np.zeros(ca.ST_N, np.int64)
halo = 0
for eng, sb, (lo, hi, w) in zip(engs, slices, wins):
    q1, f1, s1 = eng.download()
    r0, r1 = int(w.read_first), int(w.read_end)
    q[int(hb.seq_off[lo + r0]):int(hb.seq_off[lo + r1])] = q1[int(sb.seq_off[r0]):int(sb.seq_off[r1])]
    f[lo + r0:lo + r1] = f1[r0:r1]; s += s1
    halo += sb.n_bases
    eng.close()
ok = bool(np.array_equal(q, wq) and np.array_equal(f, wf) and np.array_equal(s, ws))
serial = sum(sum(v) for v in ph.values())
crit = max(ph["columns"]) + sum(ph["depth"]) + max(ph["triggers"]) + sum(ph["fold"]) + max(ph["finish"])
print(json.dumps({"workload": f"{name} x{scale} -{preset}", "bases": hb.n_bases, "windows": len(wins), "identical_to_whole_contig": ok,
                  "whole_contig_ms": round(whole_ms, 3), "windows_serial_ms": round(serial, 3),
                  "bases_uploaded_over_contig": round(halo / hb.n_bases, 4),
                  "critical_path_ms_one_device_per_window": round(crit, 3),
                  "projected_speedup": round(whole_ms / crit, 2),
                  "phase_ms": {k: [round(x, 3) for x in v] for k, v in ph.items()}}))
