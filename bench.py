#!/usr/bin/env python3
"""bench.py -- Gbases/s quality-quantized on a chr20-shaped 30x synthetic BAM (-5 preset).

One "step" = one full pass of the hot path over one contig whose raw BAM-shaped arrays are already resident in HBM:
the device-side pileup construction (crumble_hip_expand: what pileup_callback + bam_plp_auto do per read,
snp_score.c:1113-1153) followed by all kernels of crumble_hip_run.  The product runs every contig exactly once, so
both are inside the timed step; `value_kernels_only` is the run without the expansion.  N>1: one process per GPU
(torch.distributed, backend nccl == RCCL), every rank owns its own contig
(contigs are the independent unit of the path) -> weak scaling, no
data-path collective; the only collective is the final all-reduce of the
quality-histogram/statistics vector.

Prints ONE JSON line on rank 0 (see the task contract); adds
  "roofline":     dominant kernel, algorithmic bytes / HIP-event time vs 8 TB/s
  "cpu_baseline": the CPU oracle ("port") timed on a bounded prefix of the same
                  workload on this host, 1 core (reference crumble is single-threaded)
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (four by default).  The file -> file legs
# start crumble-hip, whose device pipeline runs a dozen streams and sets two for itself; with THIS process holding four more
# beside it the legs were box- and run-dependent (0.09-0.29 s of loop), with two here as well they are 0.089-0.091 s run after
# run (profiles/NOTES_r06.md).  The timed step, the streamed legs and the other configurations measure the same either way
# (174.1 / 174.1 Gbases/s, 26.5 / 27.0, 107 / 124).  Must be in the environment before the runtime starts; a caller's value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def kernel_source_sha():
    """sha256 (16 hex digits) of the kernel sources: a profiles/*_pmc_traffic.json is used for `roofline.traffic` only
    when it was measured on exactly these kernels."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "crumble_amd", "csrc")
    for f in ("dev_common.h", "kernels.h", "kcol2.h", "p0_kernels.h", "engine.hip"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def find_pmc_json():
    """the newest profiles/*_pmc_traffic.json measured on the current kernel sources, or None"""
    import glob
    sha = kernel_source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            if json.load(open(f)).get("source_sha16") == sha:
                return f
        except Exception:            # noqa: BLE001
            pass
    return None


def find_profile_json(suffix):
    """the newest profiles/*_<suffix>.json measured on the current kernel sources, or None"""
    import glob
    sha = kernel_source_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{suffix}.json")), reverse=True):
        try:
            if json.load(open(f)).get("source_sha16") == sha:
                return f
        except Exception:            # noqa: BLE001
            pass
    return None


GPU_CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md: peak engine clock
GPU_SIMDS = 256 * 4


def secondary_roofline(kernel_key, kernel_ms, n_bases):
    """The roof the column kernel actually sits under ("secondary limiter to report honestly"): instruction
    issue.  floor_ms = SQ_INSTS_VALU wave-instructions x 4 cycles (a wave64 VALU / FP64 instruction occupies its SIMD's
    issue slot for four) / (1024 SIMDs x the peak clock), from the rocprofv3 --pmc summary in profiles/ measured on these
    kernel sources (null otherwise: never a stale constant)."""
    f = find_profile_json("sq_counters")
    if not f:
        return None
    try:
        c = json.load(open(f)).get(kernel_key)
        if not c:
            return None
        floor_ms = c["SQ_INSTS_VALU"] * 4.0 / (GPU_SIMDS * GPU_CLOCK_GHZ * 1e9) * 1e3
        out = {"bound": "valu_issue", "floor_ms": round(floor_ms, 3), "frac": round(floor_ms / kernel_ms, 4) if kernel_ms > 0 else None,
               "valu_wave_insts": int(c["SQ_INSTS_VALU"]), "salu_wave_insts": int(c.get("SQ_INSTS_SALU", 0)),
               "valu_busy": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] * (c["SQ_WAVES"] / GPU_SIMDS), 4) if c.get("SQ_WAVE_CYCLES") else None,
               "wait_any": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAVE_CYCLES") else None,
               "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4) if c.get("SQ_LDS_IDX_ACTIVE") else None,
               # the consensus sums (kcol2.h): six FP64 adds per pileup entry in file order, five more for an entry that
               # disagrees with its column's key base (the reference: 5 per entry on 15 sums + 1, snp_score.c:656-683, 651)
               "fp64_adds_per_base": 6,
               "source": os.path.basename(f)}
        return out
    except Exception:                # noqa: BLE001
        return None


def counter_traffic(kernel_prefix, config=None):
    """HBM bytes per launch of a kernel from the PMC summary of these sources: 2 x FETCH_SIZE + WRITE_SIZE (the FETCH
    calibration of tools/fetch_calib.*), for the headline config (config None) or profiles' other_configs[config]."""
    f = find_profile_json("pmc_traffic")
    if not f:
        return None, None
    try:
        d = json.load(open(f))
        kk = d["kernels"] if config is None else d.get("other_configs", {}).get(config, {})
        key = [k for k in kk if k.split("<")[0] == kernel_prefix]
        if not key:
            return None, None
        t = kk[key[0]]
        return round((2.0 * t["FETCH_SIZE_KB_per_launch"] + t["WRITE_SIZE_KB_per_launch"]) * 1024), os.path.basename(f) + " (2 x FETCH_SIZE + WRITE_SIZE, per launch)"
    except Exception:                # noqa: BLE001
        return None, None


def prefix_batch(hb, n_reads):
    """First n_reads reads of a HostBatch (same data, bounded CPU-baseline sample)."""
    from crumble_amd import HostBatch
    n = min(n_reads, hb.n_reads)
    nb, nc = int(hb.seq_off[n]), int(hb.cig_off[n])
    return HostBatch(hb.tid, hb.pos[:n], hb.mapq[:n], hb.flag[:n], hb.seq_off[:n + 1],
                     hb.cig_off[:n + 1], hb.cigar[:nc], hb.bases[:nb], hb.quals[:nb])


def _host_threads():
    """Threads the BAM reader/writer uses: min(visible CPUs, affinity mask, cgroup cpu.max quota) -- the rule of
    crumble_amd/csrc/bamio.c."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except Exception:
        pass
    return n


def codec_kernels(path, device):
    """The device BGZF codec (crumble_gz_*, csrc/gz_kernels.h) alone on a BAM file's bytes: members inflated, the records deflated
    again; HIP-event time of the kernels and of the copies, sizes against the host codec's.  Algorithmic bytes of a pass =
    compressed + uncompressed (each crosses HBM once at least); the kernels are bound by latency and instruction issue, not by
    bytes."""
    import gzip
    from crumble_amd import gz
    comp = np.fromfile(path, np.uint8)
    c = gz.GzCodec(device)
    try:
        raw = c.inflate(comp)                                   # (first call: allocations)
        t0 = time.perf_counter(); raw = c.inflate(comp); w_in = time.perf_counter() - t0
        ti = c.times()
        ok_in = bool(np.array_equal(raw[:1 << 20], np.frombuffer(gzip.decompress(comp.tobytes())[:1 << 20], np.uint8)))
        out = c.deflate(raw, 6)
        t0 = time.perf_counter(); out = c.deflate(raw, 6); w_out = time.perf_counter() - t0
        td = c.times()
        back = c.inflate(out)
        ok_out = bool(np.array_equal(back, raw))
        n_members = int(len(gz.bgzf_scan(comp)[0]))
        return {"raw_bytes": int(raw.size), "members": n_members,
                "inflate": {"kernels_ms": round(ti["kernels_ms"], 2), "GB_per_s_kernels": round(raw.size / ti["kernels_ms"] / 1e6, 2),
                            "wall_ms_incl_copies": round(w_in * 1e3, 1), "h2d_ms": round(ti["h2d_ms"], 2), "d2h_ms": round(ti["d2h_ms"], 2),
                            "hbm_frac": round((raw.size + comp.size) / (ti["kernels_ms"] * 1e-3) / 8e12, 5), "equals_gzip": ok_in},
                "deflate": {"kernels_ms": round(td["kernels_ms"], 2), "GB_per_s_kernels": round(raw.size / td["kernels_ms"] / 1e6, 2),
                            "wall_ms_incl_copies": round(w_out * 1e3, 1), "h2d_ms": round(td["h2d_ms"], 2), "d2h_ms": round(td["d2h_ms"], 2),
                            "hbm_frac": round((raw.size + out.size) / (td["kernels_ms"] * 1e-3) / 8e12, 5),
                            "bytes": int(out.size), "bytes_host_level1": int(comp.size), "round_trip_ok": ok_out}}
    finally:
        c.close()


def file_to_file(ca, synth, refs, batches, preset, device, label, verify_seconds=2.0, codec=None, level="1", codec_probe=False, pipeline=None):
    """Metric (ii) of SURVEY.md 8(d): contigs as a BAM file through the crumble-hip driver (BGZF inflate + BAM parse +
    H2D + expansion + kernels + D2H + BGZF deflate).  `value` is over the driver's own clock from the first byte read
    to the last byte written (the figure crumble-hip -v prints: reader, GPU worker and writer threads overlap across
    contigs); `value_process` over the wall clock of the whole process (start, hipInit, engine creation too).  The
    output file is read back: one contig's qualities are compared with the oracle (a bounded prefix of a large one)."""
    import ctypes as C
    import re
    import subprocess
    import tempfile
    from crumble_amd import bamio
    import oracle_lib as O
    tmp = tempfile.mkdtemp(prefix="crumble_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    src, dst = os.path.join(tmp, "in.bam"), os.path.join(tmp, "out.bam")
    total = sum(b_.n_bases for b_ in batches)
    try:
        bamio.write_batches(src, refs, batches, level=1)
        probe = None
        if codec_probe:
            probe = codec_kernels(src, device)
        env = dict(os.environ, CRUMBLE_HIP_DEVICE=str(device))
        if level is not None:
            env["CRUMBLE_BAM_LEVEL"] = level          # "1": the fastest host deflate (the legs of rounds 4 and 5); None: the default level (6)
        else:
            env.pop("CRUMBLE_BAM_LEVEL", None)
        if codec:
            env["CRUMBLE_HIP_CODEC"] = codec          # (host pipeline only) gpu: BGZF members inflated / deflated on the device | host
        if pipeline:
            env["CRUMBLE_HIP_PIPELINE"] = pipeline    # device (crumble-hip's default: the records stay in HBM from inflate to deflate) | host
        lvl = [f"-{preset}"] if preset else []
        # two invocations, the one with the shorter loop reported (`runs_loop_s` has both): a process that starts while the
        # driver is still clearing the device memory of processes that just ended waits in its first allocations -- seen as
        # 0.2 s of hipMalloc inside the first contigs' loop ("device allocations inside" of the -v line) in one bench run in ten
        best, loops = None, []
        for _rep in range(2):
            t0 = time.perf_counter()
            r_ = subprocess.run([os.path.join(ROOT, "crumble_amd", "crumble-hip"), "-v"] + lvl + [src, dst],
                                env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            wall_ = time.perf_counter() - t0
            if r_.returncode:
                return {"error": r_.stderr[-300:]}
            m_ = re.search(r"end-to-end ([0-9.]+) s \(read\+parse ([0-9.]+), upload\+kernels\+download ([0-9.]+), write ([0-9.]+)\)", r_.stderr)
            st_ = [float(x) for x in m_.groups()] if m_ else [None] * 4
            loops.append(st_[0])
            if best is None or (st_[0] is not None and (best[1][0] is None or st_[0] < best[1][0])):
                best = (r_, st_, wall_)
        r, st, wall = best
        L = bamio.lib()
        L.crumble_bam_codec.restype = C.c_char_p
        out = {"workload": label, "value": round(total / st[0] / 1e9, 4) if st[0] else None, "unit": "Gbases/s",
               "loop_s": st[0], "read_parse_s": st[1], "gpu_incl_pcie_s": st[2], "deflate_write_s": st[3],
               "value_process": round(total / wall / 1e9, 4), "process_wall_s": round(wall, 2), "contigs": len(batches), "bases": total,
               "host_threads": _host_threads(), "host_cpus_visible": os.cpu_count(),
               "codec": (re.search(r"BGZF codec: (.*)", r.stderr) or [None, L.crumble_bam_codec().decode()])[1],
               "in_bytes": os.path.getsize(src), "out_bytes": os.path.getsize(dst),
               "host_stage_s": (re.search(r"bamio: (.*)", r.stderr) or [None, None])[1],
               "pipeline": (re.search(r"pipeline: (device) \((.*)\)", r.stderr) or [None, "host", None])[1],
               "runs_loop_s": loops}
        m2 = re.search(r"pipeline: device \(.*GPU stage: (.*)\)", r.stderr)
        if m2:
            out["gpu_stage_s"] = m2.group(1)
            # the device pipeline moves the compressed file over PCIe, once each way, and nothing else of size
            out["pcie_bytes_per_base"] = {"in": round(out["in_bytes"] / total, 4), "out": round(out["out_bytes"] / total, 4)}
        if probe:
            out["codec_kernels"] = probe
        if verify_seconds > 0:
            got = list(bamio.Reader(dst))
            params = ca.preset_params(preset)
            k_c = int(np.argmin([b_.n_bases for b_ in batches]))
            hb = batches[k_c]
            n_sample = int(min(hb.n_reads, max(200, verify_seconds / 60e-9 / max(1, hb.n_bases / max(1, hb.n_reads)))))
            wq, wf, _ = O.contig(params, prefix_batch(hb, n_sample))
            k = interior_reads(hb, n_sample) if n_sample < hb.n_reads else hb.n_reads
            nbk = int(hb.seq_off[k])
            ok = len(got) == len(batches) and all(g.n_reads == b_.n_reads for g, b_ in zip(got, batches))
            out["verified_output_file_reads"] = k if (ok and k > 0 and np.array_equal(got[k_c].quals[:nbk], wq[:nbk])) else False
        return out
    finally:
        for f in (src, dst):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)


def end_to_end_legs(ca, synth, shard, args, hb_full, label, device):
    """The file -> file legs of the default line: a 24-contig WGS-shaped file (BASELINE config 5's shape, scaled so that
    the leg takes seconds: the driver overlaps reading, GPU work and writing of consecutive contigs) -- `value` --
    and one chr20-shaped contig (config 2's shape: one contig, the three stages one after another)."""
    out = {}
    if args.e2e:
        out = file_to_file(ca, synth, [("chr20", 64444167)], [hb_full], args.preset, device, f"{label} x {args.scale} (one contig)")
        return out
    sizes = [int(l * 0.005 * args.scale) for l in shard.GRCH38]
    batches = []
    for tid, ln in enumerate(sizes):
        cfg = synth.config("chr20", scale=1.0)
        cfg.contig_len = max(ln, 4 * cfg.read_len + 4096)
        cfg.seed = 7000 + tid
        cfg.tid = tid
        batches.append(synth.generate(cfg))
    refs = [(f"chr{t + 1}", max(s_, 4 * 150 + 4096) + 1000) for t, s_ in enumerate(sizes)]
    out = file_to_file(ca, synth, refs, batches, args.preset, device, f"wgs-30x: 24 contigs (GRCh38 lengths x {0.005 * args.scale:g})", codec_probe=True)
    # crumble-hip's default is the device pipeline (CRUMBLE_HIP_PIPELINE=auto: the records stay in HBM from inflate to deflate,
    # csrc/stream_host.h) -- the leg above.  The same file through the host pipeline (reader / GPU workers / writer threads, the
    # records parsed and written on the host) with its codec policy (the host's threads inflate, the device deflates: round 6's
    # first figure), with the host's threads doing both (libdeflate level 1 where the box has it: the legs of rounds 4 and 5), and
    # with every member inflated AND deflated on the device:
    keys = ("value", "loop_s", "runs_loop_s", "read_parse_s", "gpu_incl_pcie_s", "deflate_write_s", "codec", "pipeline", "gpu_stage_s", "out_bytes", "host_stage_s", "error")
    hp = file_to_file(ca, synth, refs, batches, args.preset, device, "the same, CRUMBLE_HIP_PIPELINE=host", verify_seconds=0, pipeline="host")
    out["host_pipeline"] = {k: hp.get(k) for k in keys if k in hp}
    hc = file_to_file(ca, synth, refs, batches, args.preset, device, "the same, host pipeline, CRUMBLE_HIP_CODEC=host", verify_seconds=0, codec="host", pipeline="host")
    out["host_codec"] = {k: hc.get(k) for k in keys if k in hc}
    dv = file_to_file(ca, synth, refs, batches, args.preset, device, "the same, host pipeline, CRUMBLE_HIP_CODEC=gpu", verify_seconds=0, codec="gpu", pipeline="host")
    out["device_codec"] = {k: dv.get(k) for k in keys if k in dv}
    # ... and at the DEFAULT compression level (what htslib and reference crumble write: zlib 6): the default driver against the
    # host pipeline with the host's threads as the codec
    da = file_to_file(ca, synth, refs, batches, args.preset, device, "the same at the default level, auto", verify_seconds=0, level=None)
    dh = file_to_file(ca, synth, refs, batches, args.preset, device, "the same at the default level, host", verify_seconds=0, codec="host", level=None, pipeline="host")
    out["default_level"] = {"auto": {k: da.get(k) for k in keys if k in da}, "host_codec": {k: dh.get(k) for k in keys if k in dh}}
    # many small contigs (hs38DH: 3171 decoys and alts of a few kb): 600 contigs of 5 kb through both pipelines
    try:
        import copy
        pool = []
        for k in range(8):
            cfg_s = synth.config("chr20", scale=1.0)
            cfg_s.contig_len = 5000
            cfg_s.seed = 9000 + k
            pool.append(synth.generate(cfg_s))
        sb, srefs = [], []
        for tid in range(600):
            b_ = copy.copy(pool[tid % 8])
            b_.tid = tid
            sb.append(b_)
            srefs.append((f"d{tid}", 6000))
        sc = {}
        for name, pl in (("device", None), ("host", "host")):
            r_ = file_to_file(ca, synth, srefs, sb, args.preset, device, "600 contigs of 5 kb", verify_seconds=0, pipeline=pl)
            sc[name] = {"value": r_.get("value"), "loop_s": r_.get("loop_s"), "pipeline": r_.get("pipeline"),
                        "ms_per_contig": round(r_["loop_s"] / 600 * 1e3, 3) if r_.get("loop_s") else None, "error": r_.get("error")}
        out["small_contigs"] = {"workload": "600 contigs of 5 kb at 30x (8 different ones in turn)", "unit": "Gbases/s", **sc}
    except Exception as ex:
        out["small_contigs"] = {"error": repr(ex)[:300]}
    cfg_e = synth.config("chr20", scale=0.25 * args.scale)
    cfg_e.seed += 77
    out["one_contig"] = file_to_file(ca, synth, [("chr20", 64444167)], [synth.generate(cfg_e)], args.preset, device,
                                     f"chr20-30x x {0.25 * args.scale:g} ({cfg_e.contig_len} bp), one contig")
    return out


WORKLOADS = {   # name -> (label, default preset, default tech, keep-BED regions at scale 1)
    "chr20": ("chr20-30x", 5, None, 0),          # BASELINE config 2 (the 1xMI355X config the metric is quoted on)
    "panel": ("panel-300x", 9, None, 0),         # config 3
    "hifi": ("hifi-20x", None, "pbccs", 5000),   # config 4: -y pbccs -R keep.bed (5000 x 2 kb)
    "wgs": ("wgs-30x", 5, None, 0),              # config 5: 24 contigs, GRCh38 lengths
}


def keep_bed(contig_len, n_regions, width=2000):
    """n evenly spread keep-regions of `width` bp (5000 x 2 kb on chr1)."""
    if n_regions <= 0:
        return None
    starts = (np.arange(n_regions, dtype=np.int64) * (contig_len // n_regions) + 10000).astype(np.int64)
    starts = starts[starts + width < contig_len]
    return np.stack([starts, starts + width], 1).astype(np.int32)


def interior_reads(hb, n_sample):
    """Number k of leading reads of the first n_sample reads whose every column has its full pileup inside the
    sample (all reads covering a column start at or before it): the oracle run on the prefix alone gives the exact
    bytes, flags of reads 0..k-1 of the full contig."""
    last_pos = int(hb.pos[n_sample - 1])
    ops = hb.cigar[:int(hb.cig_off[n_sample])]
    cons = np.isin(ops & 15, (0, 2, 3, 7, 8)) * (ops >> 4).astype(np.int64)     # M D N = X consume the reference
    csum = np.concatenate([[0], np.cumsum(cons)])
    reflen = csum[hb.cig_off[1:n_sample + 1]] - csum[hb.cig_off[:n_sample]]
    bad = np.nonzero(hb.pos[:n_sample].astype(np.int64) + reflen > last_pos)[0]
    return int(bad[0]) if len(bad) else n_sample


def _cpu_worker(job):
    """One independent unit of the all-core CPU baseline: the oracle on a window of reads of the contig."""
    import oracle_lib as O
    import crumble_amd as ca
    hb, preset, tech = job
    O.load()
    p = ca.preset_params(preset, tech=tech)
    t0 = time.perf_counter()
    O.contig(p, hb)
    return hb.n_bases, time.perf_counter() - t0


def window_batch(hb, r0, n_reads):
    from crumble_amd import HostBatch
    r1 = min(hb.n_reads, r0 + n_reads)
    b0, b1 = int(hb.seq_off[r0]), int(hb.seq_off[r1])
    c0, c1 = int(hb.cig_off[r0]), int(hb.cig_off[r1])
    return HostBatch(hb.tid, hb.pos[r0:r1], hb.mapq[r0:r1], hb.flag[r0:r1], hb.seq_off[r0:r1 + 1] - b0,
                     hb.cig_off[r0:r1 + 1] - c0, hb.cigar[c0:c1], hb.bases[b0:b1], hb.quals[b0:b1])


def _algo_bytes(b_):
    """3 B per base + 32 B per read + 8 B per CIGAR op."""
    return 3.0 * b_.n_bases + 32.0 * b_.n_reads + 8.0 * int(b_.cig_off[-1])


def measure_config(ca, synth, workload, scale, device, steps=3, warmup=1, verify_seconds=3.0):
    """One more BASELINE configuration through the same engine, raw arrays resident, device-side expansion + kernels per
    step (what `value` is for the headline config): ms per step, the dominant kernel's and the whole path's fraction of the HBM roof, and the
    TIMED engine's output checked against the oracle on the leading reads whose columns lie inside a bounded sample."""
    import oracle_lib as O
    label, d_preset, d_tech, d_bed = WORKLOADS[workload]
    cfg = synth.config(workload, scale=scale)
    bed = keep_bed(cfg.contig_len, int(d_bed * scale + 0.5)) if d_tech == "pbccs" else None
    t0 = time.time()
    hb = synth.generate(cfg, bed=bed)
    t_gen = time.time() - t0
    if hasattr(ca, "lib"):
        hb.pack4(ca.lib()).lens()
    params = ca.preset_params(d_preset, tech=d_tech)
    eng = ca.Engine(params, device, options={"stages": "on"})
    try:
        eng.upload(hb)
        for _ in range(warmup):
            eng.expand()
            eng.run()
        kt = {}
        p0_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.expand()                                 # device-side pileup construction (P0) of the resident raw arrays
            eng.run()                                    # every kernel of the path + one stream sync
            p0_ms += eng.upload_times()[1]
            for name, ms, by in eng.kernel_times():
                a = kt.setdefault(name, [0.0, 0.0, 0])
                a[0] += ms; a[1] += by; a[2] += 1
        step_s = (time.perf_counter() - t0) / steps
        p0_ms /= steps
        dom = max(kt.items(), key=lambda kv: kv[1][0])
        dom_ms, dom_bytes = dom[1][0] / dom[1][2], dom[1][1] / dom[1][2]
        kernels_ms = sum(v[0] for v in kt.values()) / steps
        out = {
            "workload": f"{label} synthetic contig ({cfg.contig_len} bp, {hb.n_reads} reads, {hb.n_bases} bases), "
                        f"{'-' + str(d_preset) if d_preset else 'default level'}{' -y ' + d_tech if d_tech else ''}"
                        f"{' -R keep.bed (' + str(len(hb.bed_start)) + ' regions)' if len(hb.bed_start) else ''}",
            "value": round(hb.n_bases / step_s / 1e9, 3), "unit": "Gbases/s", "ms_per_step": round(step_s * 1e3, 3),
            "roofline": {"bound": "hbm", "kernel": dom[0], "kernel_ms": round(dom_ms, 3),
                         "achieved": round(dom_bytes / (dom_ms * 1e-3) / 1e9, 2) if dom_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if dom_ms > 0 else 0.0,
                         "traffic": counter_traffic(dom[0], workload)[0] if scale == 1.0 else None,
                         "traffic_source": counter_traffic(dom[0], workload)[1] if scale == 1.0 else None,
                         "secondary": secondary_roofline({"panel": "k_columns2_panel", "hifi": "k_columns2_hifi"}.get(workload, ""), dom_ms, hb.n_bases)
                                      if (scale == 1.0 and dom[0] == "k_columns2") else None},
            "pipeline_frac": round(_algo_bytes(hb) / ((kernels_ms + p0_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "value_kernels_only": round(hb.n_bases / (kernels_ms * 1e-3) / 1e9, 3), "p0_ms": round(p0_ms, 3),
            "kernels_ms": {k: round(v[0] / v[2], 3) for k, v in kt.items()}, "gen_s": round(t_gen, 1),
        }
        if verify_seconds > 0:
            tq, tf, _ = eng.download(want_stats=False)
            n_sample = int(min(hb.n_reads, max(200, verify_seconds / 60e-9 / max(1, hb.n_bases / max(1, hb.n_reads)))))
            sample = prefix_batch(hb, n_sample)
            if len(hb.bed_start):
                sample.bed_start, sample.bed_end = hb.bed_start, hb.bed_end
            wq, wf, _ = O.contig(params, sample)
            k = interior_reads(hb, n_sample)
            nbk = int(hb.seq_off[k])
            out["verified_timed_output_leading_reads"] = k if (k > 0 and np.array_equal(tq[:nbk], wq[:nbk]) and
                                                               np.array_equal(tf[:k], wf[:k])) else False
        return out
    finally:
        eng.close()


def measure_small_contig(ca, synth, device, preset=5, contig_len=5000, reps=200):
    """What a contig costs beyond its bases (snp_score.c:1478-1488: all a contig boundary costs the reference is a reset):
    crumble_hip_contig() -- copy in, device-side expansion, run, copy out -- on a 5 kb contig at 30x, repeated; and
    the run alone.  Wall clock per call."""
    cfg = synth.config("chr20", scale=1.0)
    cfg.contig_len = contig_len
    hb = synth.generate(cfg)
    eng = ca.Engine(ca.preset_params(preset), device)
    try:
        for _ in range(5):
            eng.contig(hb)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.contig(hb)
        t_contig = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.run()
        t_run = (time.perf_counter() - t0) / reps
    finally:
        eng.close()
    return {"contig_bp": contig_len, "reads": hb.n_reads, "bases": hb.n_bases, "contig_ms": round(t_contig * 1e3, 4),
            "run_ms": round(t_run * 1e3, 4), "reps": reps,
            "note": "crumble_hip_contig() per call on a 5 kb contig (host -> device, expansion, run, device -> host), and crumble_hip_run() alone"}


def measure_wgs_streamed(ca, synth, shard, scale, device, preset=5, n_engines=4, coded=True, n_uploaders=1, coded_in=None):
    """BASELINE config 5 on ONE GPU as the stream it is (snp_score.c:1437: contigs one after another): the 24 contigs
    sit in host memory; contig k+1's host -> device copy + device-side expansion, contig k's kernels, contig k-1's
    coded download and contig k-2's expansion on the host run side by side -- four host threads, a ring of engines
    (each its own HIP stream).  coded_in: a fifth stage in front packs contig k+2's qualities into 2- / 4-bit codes
    (crumble_pack_quals on the host's threads, inside the clock) and the upload sends those (crumble_batch.qualsp).  Reports the wall time per pass, the stages' summed times and wall / sum
    (1.0 = nothing overlaps)."""
    import queue
    import threading
    sizes = [int(l * scale) for l in shard.GRCH38]
    batches = []
    t0 = time.time()
    for tid, ln in enumerate(sizes):
        cfg = synth.config("chr20", scale=1.0)
        cfg.contig_len = max(ln, 4 * cfg.read_len + 4096)
        cfg.seed = 5000 + tid
        cfg.tid = tid
        batches.append(synth.generate(cfg))
        if hasattr(ca, "lib"):
            batches[-1].pack4(ca.lib()).lens()      # bases as BAM packs them, BAM's per-read lengths: what crumble-hip sends
    t_gen = time.time() - t0
    params = ca.preset_params(preset)
    n_engines = int(os.environ.get("CRUMBLE_BENCH_STREAM_ENGINES", n_engines))
    eopts = {"dl_kernel": os.environ["CRUMBLE_BENCH_DL_KERNEL"]} if os.environ.get("CRUMBLE_BENCH_DL_KERNEL") else None
    engines = [ca.Engine(params, device, options=eopts) for _ in range(n_engines)]
    # what crosses PCIe lives in page-locked host memory (crumble_hip_host_alloc), as a host that keeps its buffers would
    # hold it: the batches' arrays (copied there once, before the clock) and one output buffer pair per engine
    pins = []

    def pin(a):
        b, p_ = ca.pinned_array(len(a), a.dtype)
        b[:] = a
        pins.append(p_)
        return b
    originals = []                                  # (restored before the page-locked blocks are freed)
    for b_ in batches:
        saved = {}
        for nm in ("pos", "mapq", "flag", "seq_off", "cig_off", "cigar", "quals") + (("l_qseq", "n_cigar") if getattr(b_, "l_qseq", None) is not None else ()):
            saved[nm] = getattr(b_, nm)
            setattr(b_, nm, pin(saved[nm]))
        if getattr(b_, "bases4", None) is not None:
            saved["bases4"] = b_.bases4
            b_.bases4 = pin(b_.bases4)
        originals.append(saved)
    max_nb, max_n = max(b_.n_bases for b_ in batches), max(b_.n_reads for b_ in batches)
    out_bufs = {id(e): (pin(np.zeros(max(1, max_nb), np.uint8)), pin(np.zeros(max(1, max_n), np.uint8))) for e in engines}
    # the coded download's three streams (2 bits per base, a block table, the escape bytes) land in page-locked memory too
    coded_bufs = {id(e): (pin(np.zeros(max_nb // 16 + 2, np.uint32)), pin(np.zeros(max_nb // 4096 + 2, np.int64)),
                          pin(np.zeros(max(1, max_nb // 8), np.uint8))) for e in engines} if coded else {}
    # the coded qualities are made INSIDE the pass (a stage of its own); their page-locked landing buffers are made here
    import ctypes as C_
    if coded_in is None:        # off by default: on a 16-core host the packing costs more CPU than the link saves (profiles/NOTES_r05.md)
        coded_in = os.environ.get("CRUMBLE_BENCH_STREAM_CODED_IN") == "1"
    coded_in = bool(coded_in and hasattr(ca, "lib") and hasattr(ca.lib(), "crumble_pack_quals"))
    qp_bufs = [pin(np.zeros((b_.n_bases * 4 + 7) // 8 + 16, np.uint8)) for b_ in batches] if coded_in else []
    if coded_in:
        ca.lib().crumble_pack_quals.restype = C_.c_int
        ca.lib().crumble_pack_quals.argtypes = [C_.c_int64, C_.c_void_p, C_.c_void_p, C_.c_void_p]
    # largest contig first: what is left when the last copy has landed (that contig's expansion, run, download, host expansion)
    # is then the smallest contig's
    order = sorted(range(len(batches)), key=lambda i: -batches[i].n_bases)
    split_upload = hasattr(engines[0], "upload_raw") and os.environ.get("CRUMBLE_BENCH_STREAM_SPLIT", "1") == "1"
    xfer = [0, 0]
    k_small = int(np.argmin([b_.n_bases for b_ in batches]))
    kept = {}
    lens_ok = [False] * len(batches)
    try:
        def one_pass(keep):
            free_q, run_q, dl_q, dec_q = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
            for e in engines:
                free_q.put(e)
            t_stage = [0.0, 0.0, 0.0, 0.0, 0.0]
            err = []
            up_q = queue.Queue()

            up_lock = threading.Lock()
            up_state = {"live": n_uploaders}

            def packer():
                # the qualities as 2- / 4-bit codes (crumble_pack_quals, the host's threads): what the upload then sends
                try:
                    for i in order:
                        b_ = batches[i]
                        if coded_in:
                            t = time.perf_counter()
                            lut = np.zeros(16, np.uint8)
                            bits = ca.lib().crumble_pack_quals(b_.n_bases, b_.quals.ctypes.data, qp_bufs[i].ctypes.data, lut.ctypes.data)
                            if bits:
                                b_.qualsp, b_.qual_bits, b_.qual_lut = qp_bufs[i][:(b_.n_bases * bits + 7) // 8 + 16], bits, lut
                            else:
                                b_.qualsp = None
                            t_stage[4] += time.perf_counter() - t
                        up_q.put(i)
                except Exception as ex:            # noqa: BLE001
                    err.append(ex)
                for _ in range(n_uploaders):
                    up_q.put(None)

            def uploader():
                # n_uploaders threads take the contigs in turn: while one engine's expansion (a few small kernels and their
                # size read-back) runs, the other's copies keep the link busy
                try:
                    while True:
                        i = up_q.get()
                        if i is None:
                            break
                        e = free_q.get()
                        # the copy alone (crumble_hip_upload_raw returns when the arrays are on the device): the expansion is
                        # the next stage's, so that the link goes straight on with the next contig
                        t = time.perf_counter(); (e.upload_raw if split_upload else e.upload)(batches[i]); dt = time.perf_counter() - t
                        with up_lock:
                            t_stage[0] += dt
                        run_q.put((i, e))
                except Exception as ex:            # noqa: BLE001
                    err.append(ex)
                with up_lock:
                    up_state["live"] -= 1
                    last = up_state["live"] == 0
                if last:
                    run_q.put(None)

            def runner():
                try:
                    while True:
                        it = run_q.get()
                        if it is None:
                            break
                        t = time.perf_counter()
                        if split_upload:
                            it[1].expand()
                        it[1].run()
                        t_stage[1] += time.perf_counter() - t
                        dl_q.put(it)
                except Exception as ex:            # noqa: BLE001
                    err.append(ex)
                dl_q.put(None)

            def downloader():
                try:
                    while True:
                        it = dl_q.get()
                        if it is None:
                            break
                        t = time.perf_counter()
                        if coded:
                            # 2 bits per base + escapes over the link; the expansion is the next stage's
                            c_, f, _ = it[1].download_coded(want_stats=False, bufs=coded_bufs[id(it[1])])
                            q = None
                        else:
                            q, f, _ = it[1].download(want_stats=False, out=out_bufs[id(it[1])])
                        t_stage[2] += time.perf_counter() - t
                        if keep:
                            u_, d_ = it[1].transfer_bytes()
                            xfer[0] += u_; xfer[1] += d_
                        if coded:
                            dec_q.put((it[0], it[1], c_, f))
                            continue
                        if keep:
                            lens_ok[it[0]] = len(q) == batches[it[0]].n_bases
                            if it[0] == k_small:
                                kept["q"], kept["f"] = q.copy(), f.copy()
                        free_q.put(it[1])
                except Exception as ex:            # noqa: BLE001
                    err.append(ex)
                dec_q.put(None)

            def decoder():
                # the coded result expanded on the host against the input qualities it still holds (crumble_decode_quals
                # on the host's threads): the bytes a plain download gives
                try:
                    while True:
                        it = dec_q.get()
                        if it is None:
                            break
                        t = time.perf_counter()
                        q = (it[1].decode(it[2], batches[it[0]].quals, out=out_bufs[id(it[1])][0], threads=int(os.environ.get("CRUMBLE_BENCH_DECODE_THREADS", "0")))
                             if not os.environ.get("CRUMBLE_BENCH_STREAM_NODECODE") else out_bufs[id(it[1])][0][:batches[it[0]].n_bases])
                        t_stage[3] += time.perf_counter() - t
                        if keep:
                            lens_ok[it[0]] = len(q) == batches[it[0]].n_bases
                            if it[0] == k_small:
                                kept["q"], kept["f"] = q.copy(), it[3].copy()
                        free_q.put(it[1])
                except Exception as ex:            # noqa: BLE001
                    err.append(ex)
            th = [threading.Thread(target=f) for f in (packer,) + (uploader,) * n_uploaders + (runner, downloader) + ((decoder,) if coded else ())]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            wall = time.perf_counter() - t0
            if err:
                raise err[0]
            return wall, t_stage
        one_pass(False)                                   # warm-up: code objects, buffer growth, page-locked buffers
        wall, st = one_pass(True)
        total = sum(b.n_bases for b in batches)
        # the smallest contig's bytes and flags against the oracle (bounded CPU time), everything else by properties
        import oracle_lib as O
        wq, wf, _ = O.contig(params, batches[k_small])
        ok = bool(np.array_equal(kept["q"], wq) and np.array_equal(kept["f"], wf)) and all(lens_ok)
        return {
            "workload": f"wgs-30x: 24 synthetic contigs (GRCh38 lengths x {scale}), {total} bases, -{preset}, streamed through "
                        f"{n_engines} engines on one GPU ({'host packing || ' if coded_in else ''}{n_uploaders} uploads || kernels || download || host expansion)",
            "value_incl_pcie": round(total / wall / 1e9, 3), "unit": "Gbases/s", "ms_per_pass": round(wall * 1e3, 2),
            "stages_ms": {("h2d" if split_upload else "h2d_and_p0"): round(st[0] * 1e3, 2), ("p0_and_kernels" if split_upload else "kernels"): round(st[1] * 1e3, 2), "d2h": round(st[2] * 1e3, 2),
                          **({"host_decode": round(st[3] * 1e3, 2)} if coded else {}), **({"host_pack": round(st[4] * 1e3, 2)} if coded_in else {})},
            "upload": ("coded qualities: 2 / 4 bits per base into a table of the batch's values (crumble_pack_quals on the host's threads, "
                       "a stage inside the clock; expanded on the device by k_p0_quals), BAM-packed bases, per-read lengths"
                       if coded_in else "one byte per quality, BAM-packed bases, per-read lengths"),
            "download": "coded: 2 bits per base + escape bytes (crumble_hip_encode / _download_coded), expanded by crumble_decode_quals on the host's threads as a fourth stage" if coded else "plain: one byte per base",
            "pcie_bytes_per_base": {"in": round(xfer[0] / max(1, total), 4), "out": round(xfer[1] / max(1, total), 4)},
            "overlap_wall_over_sum_of_stages": round(wall / max(1e-9, sum(st)), 3),
            "verified_smallest_contig": ok, "gen_s": round(t_gen, 1),
            "note": "batches and output buffers in page-locked host memory (crumble_hip_host_alloc); a PCIe-inclusive rate, never `value`",
        }
    finally:
        for e in engines:
            e.close()
        for b_, saved in zip(batches, originals):   # no array of a batch may point into freed page-locked memory
            for nm, a in saved.items():
                setattr(b_, nm, a)
            b_.qualsp = None
        qp_bufs.clear()
        out_bufs.clear()
        coded_bufs.clear()
        kept.clear()
        for p_ in pins:
            ca.pinned_free(p_)


def allreduce_stats(ca, engines, dist, rank, world, red_dev):
    """The one collective of the path: the statistics / quality-histogram vector summed over the ranks.  A rank that ran
    several engines (one per contig) first folds their vectors into its first engine (crumble_hip_stats_add), then the
    product's own collective runs on that engine: crumble_hip_stats_allreduce (RCCL, in place on the device vector)
    over a communicator built from an id that rank 0 hands round.  Should the communicator fail to come up on some box
    (or the ranks share one device: the gloo test), the same sum goes through torch.distributed instead of losing the
    run.  -> (summed vector, how it was summed)"""
    import torch
    eng = engines[0]
    if len(engines) > 1:
        rest = sum(e_.stats_read() for e_ in engines[1:])
        eng.stats_add(rest)
        for e_ in engines[1:]:
            e_.stats_reset()
    stats = eng.stats_read()
    if dist is None:
        return stats, "none (one rank)"
    done = False
    stats_via = None
    if red_dev == "cuda":
        import ctypes as C
        ok, comm, rccl = 1, None, None

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid = UniqueId()
        try:
            rccl = C.CDLL("librccl.so")
            if rank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
                ok = 0
        except Exception:                       # noqa: BLE001
            ok = 0
        # rank 0 ALWAYS broadcasts (the id, or None when it has none): no rank is left waiting in the broadcast
        box = [bytes(uid.internal) if (rank == 0 and ok) else None]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            ok = 0
        if ok:
            try:
                C.memmove(C.byref(uid), box[0], 128)
                comm = C.c_void_p()
                rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
                ok = 1 if rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0 else 0
                if not ok:
                    comm = None
            except Exception:                   # noqa: BLE001
                ok, comm = 0, None
        allok = torch.tensor([ok], dtype=torch.int32, device=red_dev)
        dist.all_reduce(allok, op=dist.ReduceOp.MIN)            # every rank takes the same path
        if int(allok.item()) == 1:
            rc = eng.L.crumble_hip_stats_allreduce(eng.h, comm)
            # (a refusal -- e.g. the loaded librccl's version against the build's rccl.h -- comes before any
            # collective is issued and is the same on every rank: agree on it, then take the other path)
            allok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(allok, op=dist.ReduceOp.MIN)
            if int(allok.item()) == 1:
                stats = eng.stats_read()
                stats_via = "crumble_hip_stats_allreduce (RCCL)"
                done = True
            elif rank == 0:
                print("bench: crumble_hip_stats_allreduce refused (%s); statistics summed by torch.distributed"
                      % eng.L.crumble_hip_last_error().decode(), file=sys.stderr)
        if comm is not None and rccl is not None:           # (also on the ranks whose communicator came up while another's did not)
            rccl.ncclCommDestroy.argtypes = [C.c_void_p]
            rccl.ncclCommDestroy(comm)
    if not done:
        st = torch.from_numpy(stats.copy()).to(red_dev)
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
        stats = st.cpu().numpy()
        stats_via = f"torch.distributed all_reduce ({'RCCL' if red_dev == 'cuda' else 'gloo'})"
    return stats, stats_via


def measure_wgs_strong(ca, synth, shard, dist, rank, world, local_rank, red_dev, scale, preset=5, steps=2, warmup=1):
    """BASELINE config 5 at N > 1 (what the metric is named after): 24 synthetic contigs with GRCh38's proportions,
    LPT-packed onto the ranks by length (contigs are the independent unit: every cross-column state resets on a tid change,
    snp_score.c:1478-1488) -- strong scaling, no data-path collective -- resident in HBM, `steps` passes of expansion + run
    over each rank's share, max over ranks; then ONE all-reduce of the statistics vector through the product's
    crumble_hip_stats_allreduce, and the summed histogram checked against the job's base count."""
    import torch
    sizes = [int(l * scale) for l in shard.GRCH38]
    assign, loads = shard.lpt_pack(sizes, world)
    batches = []
    for tid in assign[rank]:
        cfg = synth.config("chr20", scale=1.0)
        cfg.contig_len = max(sizes[tid], 4 * cfg.read_len + 4096)
        cfg.seed = 5000 + tid
        cfg.tid = tid
        batches.append(synth.generate(cfg).pack4(ca.lib()).lens())
    params = ca.preset_params(preset)
    engines = []
    try:
        for b_ in batches:
            e_ = ca.Engine(params, local_rank)
            e_.upload(b_)
            engines.append(e_)
        if not engines:                                  # (more ranks than contigs: this rank only takes part in the collectives)
            engines.append(ca.Engine(params, local_rank))

        def sync_all():
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        for _ in range(warmup):
            for e_ in engines[:len(batches)]:
                e_.expand(); e_.run()
        for e_ in engines:
            e_.stats_reset()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            for e_ in engines[:len(batches)]:
                e_.expand(); e_.run()
        sync_all()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        stats, via = allreduce_stats(ca, engines, dist, rank, world, red_dev)
        tb = torch.tensor([sum(b_.n_bases for b_ in batches), sum(b_.n_reads for b_ in batches)], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases, total_reads = int(tb[0].item()), int(tb[1].item())
        names = ca.capi.ST_NAMES
        hist_ok = (int(stats[ca.capi.ST_HIST0:ca.capi.ST_HIST0 + 256].sum()) == total_bases * steps and
                   int(stats[names.index("bases")]) == total_bases * steps and int(stats[names.index("reads")]) == total_reads * steps)
        return {
            "workload": f"wgs-30x: 24 synthetic contigs (GRCh38 lengths x {scale}), {total_bases} bases, -{preset}, LPT-packed on {world} GPUs",
            "value": round(total_bases * steps / elapsed / 1e9, 4), "unit": "Gbases/s", "scaling": "strong", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "lpt_max_over_mean_load": round(max(loads) / (sum(loads) / len(loads)), 4),
            "contigs_per_rank": [len(a_) for a_ in assign], "stats_allreduce": via, "summed_histogram_ok": bool(hist_ok),
        }
    finally:
        for e_ in engines:
            e_.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="chr20", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="contig length scale (1.0 = BASELINE config)")
    ap.add_argument("--preset", type=int, default=None, help="crumble -[1-9] (default: the workload's)")
    ap.add_argument("--tech", default=None, help="crumble -y (default: the workload's; 'none' disables)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--cpu-all", action="store_true", help="(default now) also time the CPU restatement on all host "
                    "cores (independent windows of the contig, one process per core): 'cpu_baseline_all'")
    ap.add_argument("--no-cpu-all", action="store_true", help="skip the all-core CPU leg")
    ap.add_argument("--contig-threads", type=int, default=4, help="wgs: host threads that run the contigs' engines "
                    "(one HIP stream each) side by side on the GPU; 1 = one contig at a time")
    ap.add_argument("--pmc-json", default=None, help="profiles/*_pmc_traffic.json measured for THIS build and "
                    "workload: fills roofline.traffic (default: the newest one in profiles/ whose source_sha16 equals the "
                    "current kernel sources' -- null when there is none)")
    ap.add_argument("--kcol", default="auto", help="crumble_hip_set_option kcol (auto|v1|v1perm|v2)")
    ap.add_argument("--pblock", default="auto", help="crumble_hip_set_option pblock (auto|reads|segments|words)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the `other_configs` object (BASELINE configs "
                    "3, 4 and 5 measured after the headline config; default line, one GPU only; --no-cpu skips it too: its "
                    "legs check themselves against the CPU oracle)")
    ap.add_argument("--wgs-strong-scale", type=float, default=0.05, help="N > 1: the `wgs_strong` object's contig length scale "
                    "(24 GRCh38-proportioned contigs, LPT-packed over the ranks, resident in HBM; times --scale)")
    ap.add_argument("--wgs-stream-scale", type=float, default=0.05, help="other_configs.wgs_streamed: contig length scale")
    ap.add_argument("--e2e", action="store_true",
                    help="metric (ii), BAM file -> crumble-hip -> BAM file (SURVEY.md 8d), on the WHOLE contig (the default line "
                         "runs it on a quarter of it); an extra 'end_to_end' object, never part of 'value'")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end leg")
    ap.add_argument("--only-streamed", action="store_true", help="tuning: run the config-5 streamed leg alone and print its object")
    ap.add_argument("--profile", action="store_true", help="the timed steps and nothing else (rocprofv3 runs: no CPU legs, no "
                    "verification, no small-contig probe, no end_to_end, no other_configs -- every launch in the trace is a step's)")
    args = ap.parse_args()
    if not args.profile and any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        # under a profiler every launch should be a timed step's, and no child process should inherit the preload
        print("bench: profiler preload detected: running as --profile (timed steps only)", file=sys.stderr)
        args.profile = True
    if args.profile:
        args.no_cpu = args.no_verify = args.no_e2e = args.no_other_configs = True
    if args.only_streamed:
        import crumble_amd as ca
        import synth
        from crumble_amd import shard
        print(json.dumps(measure_wgs_streamed(ca, synth, shard, args.wgs_stream_scale, 0)))
        return
    label, d_preset, d_tech, d_bed = WORKLOADS[args.workload]
    preset = args.preset if args.preset is not None else d_preset
    tech = (None if args.tech == "none" else args.tech) if args.tech is not None else d_tech
    if args.preset is not None and args.tech is None:
        tech = None                                   # an explicit numeric preset alone means just that preset
    args.preset = preset

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl") over xGMI is the real path; CRUMBLE_BENCH_BACKEND=gloo + CRUMBLE_BENCH_DEVICE=0 let the
        # N>1 control flow be exercised on a one-GPU box (tests/test_gpu_parity.py::test_bench_two_ranks)
        backend = os.environ.get("CRUMBLE_BENCH_BACKEND", "nccl")
        if "CRUMBLE_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["CRUMBLE_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if os.environ.get("CRUMBLE_BENCH_BACKEND", "nccl") == "nccl" else "cpu"

    import crumble_amd as ca
    import synth
    if not os.path.exists(ca.LIB_PATH):
        ca.build_library()

    # ---- synthetic input (no NA12878 BAM offline) ----
    #  chr20 (default, BASELINE config 2): one chr20-shaped contig per rank -> weak scaling
    #  wgs   (BASELINE config 5): 24 contigs with GRCh38 lengths x scale, LPT-packed onto the
    #        ranks (crumble_amd/shard.py) -> strong scaling; no data-path collective either way
    from crumble_amd import shard
    t0 = time.time()
    batches = []
    if args.workload == "wgs":
        sizes = [int(l * args.scale) for l in shard.GRCH38]
        assign, loads = shard.lpt_pack(sizes, world)
        lpt_imbalance = max(loads) / (sum(loads) / len(loads))   # max / mean of the ranks' loads: what bounds the strong scaling
        for tid in assign[rank]:
            cfg = synth.config("chr20", scale=1.0)
            cfg.contig_len = max(sizes[tid], 4 * cfg.read_len + 4096)
            cfg.seed = 5000 + tid
            cfg.tid = tid
            batches.append(synth.generate(cfg))
        scaling = "strong"
    else:
        cfg = synth.config(args.workload, scale=args.scale)
        cfg.seed = cfg.seed + 1000 * rank       # rank r owns "contig r"
        cfg.tid = rank
        bed = keep_bed(cfg.contig_len, int(d_bed * args.scale + 0.5)) if tech == "pbccs" else None
        batches.append(synth.generate(cfg, bed=bed))
        scaling = "weak"
    t_gen = time.time() - t0
    for b_ in batches:
        b_.pack4(ca.lib()).lens()   # the bases cross PCIe as BAM packs them (crumble_batch.bases4) and the offset tables as BAM's own
                                    # per-read lengths (l_qseq / n_cigar), as crumble-hip sends them
    hb = batches[0]
    params = ca.preset_params(args.preset, tech=tech)
    opts = {"stages": "on"}                     # HIP events around the run's stages: roofline.kernels_ms
    if args.kcol != "auto":
        opts["kcol"] = args.kcol
    if args.pblock != "auto":
        opts["pblock"] = args.pblock

    t0 = time.time()
    engines = []
    h2d_ms = p0_ms = 0.0
    for b_ in batches:
        e_ = ca.Engine(params, local_rank, options=opts)      # raises if no GPU (no CPU fallback)
        e_.upload(b_)                           # contigs stay resident in HBM
        a_, b2_ = e_.upload_times()
        if len(batches) == 1:
            # a process's first upload also pays the lazy load of every P0 kernel's code object and the first touch of
            # the engine's buffers (seen anywhere between 9 and 200 ms on a fresh box): what every later contig
            # pays is the second upload of the same batch
            e_.upload(b_)
            a_, b2_ = e_.upload_times()
        h2d_ms += a_; p0_ms += b2_
        engines.append(e_)
    eng = engines[0]
    t_up = time.time() - t0

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Several contigs per GPU (wgs): every contig has its own engine, i.e. its own HIP stream; a few host threads
    # drive them side by side the way `crumble-hip -j N` with repeated devices does, so one contig's launch gaps,
    # its mid-run size read-back and its tail run under another contig's kernels (~0.5 ms of each per contig).
    pool = None
    if len(engines) > 1 and args.contig_threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=args.contig_threads)

    def step_one(e_):
        e_.expand()                              # device-side pileup construction (P0) on the resident raw arrays
        e_.run()                                 # launches every kernel + stream sync

    def run_all():
        if pool is None:
            for e_ in engines:
                step_one(e_)
        else:
            list(pool.map(step_one, engines))    # (ctypes releases the GIL for the calls)

    for _ in range(args.warmup):
        run_all()
    for e_ in engines:
        e_.stats_reset()
    sync_all()
    t0 = time.perf_counter()
    ktimes = {}
    p0_step_ms = 0.0
    for _ in range(args.steps):
        run_all()
        for e_ in engines:
            p0_step_ms += e_.upload_times()[1]
            for name, ms, by in e_.kernel_times():
                a = ktimes.setdefault(name, [0.0, 0.0, 0])
                a[0] += ms; a[1] += by; a[2] += 1
    sync_all()
    if pool is not None:
        pool.shutdown()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the one collective of the path: statistics / quality histogram ----
    stats, stats_via = allreduce_stats(ca, engines, dist, rank, world, red_dev)
    bases_local = sum(b_.n_bases for b_ in batches)
    if dist is not None:
        tb = torch.tensor([bases_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = int(tb.item())
    else:
        total_bases = bases_local
    assert int(stats[ca.capi.ST_NAMES.index("bases")]) == total_bases * args.steps, "stats all-reduce mismatch"

    value = total_bases * args.steps / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3

    up_bytes = sum(e_.transfer_bytes()[0] for e_ in engines)
    # ---- the same contig with its qualities sent as codes (crumble_batch.qualsp: 2 / 4 bits per base into a table of the
    # batch's values, made by crumble_pack_quals on the host's threads, expanded on the device by k_p0_quals inside the
    # expansion): what the PCIe-inclusive rates below use when it is the faster way in.  Never `value`: the headline keeps
    # one byte per quality in HBM, as rounds 1-4 measured it.
    coded_q = None
    if len(batches) == 1 and world == 1 and not args.profile and hasattr(ca.lib(), "crumble_pack_quals"):
        import copy
        t_pack = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            hq = copy.copy(hb).packq(ca.lib())
            t_pack = min(t_pack, time.perf_counter() - t0)
        if getattr(hq, "qualsp", None) is not None:
            eng.upload(hq)
            eng.upload(hq)
            hq_h2d_ms, _ = eng.upload_times()
            hq_up_bytes = eng.transfer_bytes()[0]
            step_one(eng)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hq_p0 = 0.0
            for _ in range(args.steps):
                step_one(eng)
                hq_p0 += eng.upload_times()[1]
            torch.cuda.synchronize()
            hq_step_s = (time.perf_counter() - t0) / args.steps
            coded_q = {"bits": int(hq.qual_bits), "distinct_values_in_first_4M": int(len(np.unique(hb.quals[:4000000]))),
                       "host_pack_ms": round(t_pack * 1e3, 3), "h2d_ms": round(hq_h2d_ms, 3), "p0_ms": round(hq_p0 / args.steps, 3),
                       "ms_per_step": round(hq_step_s * 1e3, 3), "value": round(hb.n_bases / hq_step_s / 1e9, 3),
                       "pcie_in_bytes_per_base": round(hq_up_bytes / max(1, hb.n_bases), 4)}
        del hq

    # ---- download of the TIMED engine's result (what the verification below looks at) ----
    t0 = time.perf_counter()
    tq, tf, _ = eng.download(want_stats=False)
    t_down_first = time.perf_counter() - t0
    t_down_plain = 1e9
    for _ in range(3):                                             # steady state: the caller's buffers are already mapped
        t0 = time.perf_counter()
        tq, tf, _ = eng.download(want_stats=False, out=(tq, tf))
        t_down_plain = min(t_down_plain, time.perf_counter() - t0)
    down_plain_bytes = eng.transfer_bytes()[1]
    # the coded download (2 bits per base + escape bytes, include/crumble_hip.h) and its expansion on the host against the
    # input qualities: the same bytes (checked), a quarter of the link traffic
    t_down, t_dec, cbufs, cq = 1e9, 1e9, None, np.empty(max(1, hb.n_bases), np.uint8)
    for _ in range(1 if args.profile else 3):                      # (--profile: once, so that the object below exists; its launches are not a step's)
        t0 = time.perf_counter()
        c_, _, _ = eng.download_coded(want_flags=True, want_stats=False, bufs=cbufs)
        t1 = time.perf_counter()
        cq = eng.decode(c_, hb.quals, out=cq if len(cq) >= hb.n_bases else None)
        t2 = time.perf_counter()
        cbufs = c_._keep
        if t2 - t0 < t_down + t_dec:
            t_down, t_dec = t1 - t0, t2 - t1
    down_coded_bytes = eng.transfer_bytes()[1]
    coded_ok = bool(np.array_equal(cq[:hb.n_bases], tq))
    coded_escapes = int(c_.n_escapes)
    del cq, cbufs, c_

    # ---- roofline of the dominant kernel (HIP events on the engine's stream) ----
    dom = max(ktimes.items(), key=lambda kv: kv[1][0])
    dom_ms = dom[1][0] / dom[1][2]
    dom_bytes = dom[1][1] / dom[1][2]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    algo_total = sum(_algo_bytes(b_) for b_ in batches)
    kernel_ms_total = sum(v[0] for v in ktimes.values()) / args.steps
    p0_step_ms /= args.steps                                           # per step, summed over this rank's contigs
    # HBM traffic of that kernel from rocprofv3 PMC counters (FETCH_SIZE + WRITE_SIZE, separate passes): only when a
    # summary measured for THIS build and workload is named on the command line; never a stale constant
    traffic, traffic_src = None, None
    pmc_json = args.pmc_json or (find_pmc_json() if args.workload == "chr20" and args.scale == 1.0 else None)
    if pmc_json:
        try:
            kk = json.load(open(pmc_json))["kernels"]
            key = [k for k in kk if k.split("<")[0] == dom[0]]
            if key:
                t = kk[key[0]]
                # bytes moved: FETCH_SIZE counts a 128-byte request as 64 (calibrated: tools/fetch_calib.*), WRITE_SIZE is exact
                traffic = round((2.0 * t["FETCH_SIZE_KB_per_launch"] + t["WRITE_SIZE_KB_per_launch"]) * 1024)
                traffic_src = os.path.basename(pmc_json) + " (2 x FETCH_SIZE + WRITE_SIZE, per launch)"
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
        "kernel_ms": round(dom_ms, 3),
        "pipeline_achieved": round(algo_total / ((kernel_ms_total + p0_step_ms) * 1e-3) / 1e9, 2),
        "pipeline_frac": round(algo_total / ((kernel_ms_total + p0_step_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "pipeline_note": "SURVEY 8(d) bytes of the whole path / HIP-event time of expansion (P0) + run kernels",
        "kernels_ms": {"p0_expand": round(p0_step_ms, 3), **{k: round(v[0] / v[2], 3) for k, v in ktimes.items()}},
        "secondary": secondary_roofline("k_columns2", dom_ms, hb.n_bases)
                     if (dom[0] == "k_columns2" and args.workload == "chr20" and args.scale == 1.0 and args.preset == 5) else None,
    }

    # what a real contig pays around the timed step: PCIe in and out
    step_s = elapsed / args.steps
    h2d_s = h2d_ms * 1e-3
    # the way in with coded qualities, everything counted: packing on the host + the smaller copy + the step with k_p0_quals inside
    in_coded_s = (coded_q["host_pack_ms"] + coded_q["h2d_ms"] + coded_q["ms_per_step"]) * 1e-3 if coded_q else 1e9
    extra = {
        "p0_ms": round(p0_step_ms, 3), "h2d_ms": round(h2d_ms, 3),
        "d2h_ms": round((t_down + t_dec) * 1e3 * len(batches), 3), "d2h_coded_copy_ms": round(t_down * 1e3 * len(batches), 3),
        "d2h_host_decode_ms": round(t_dec * 1e3 * len(batches), 3), "d2h_plain_ms": round(t_down_plain * 1e3 * len(batches), 3),
        "d2h_first_ms": round(t_down_first * 1e3 * len(batches), 3),
        "coded_download_equals_plain": coded_ok, "coded_escape_bytes": coded_escapes,
        "pcie_bytes_per_base": {"in": round(up_bytes / max(1, bases_local), 4), "out": round(down_coded_bytes / max(1, hb.n_bases), 4),
                                "out_plain": round(down_plain_bytes / max(1, hb.n_bases), 4)},
        "value_kernels_only": round(bases_local / (kernel_ms_total * 1e-3) / 1e9, 3) if kernel_ms_total > 0 else None,
        "value_incl_pcie": round(bases_local / (min(step_s + h2d_s, in_coded_s) + min(t_down + t_dec, t_down_plain) * len(batches)) / 1e9, 3),
        "value_incl_pcie_upload": "coded qualities (host packing inside)" if in_coded_s < step_s + h2d_s else "plain",
        "value_incl_pcie_download": "coded" if t_down + t_dec < t_down_plain else "plain",
        "coded_qualities": coded_q,
        "value_incl_pcie_coded_both_copies_only": (round(bases_local / (coded_q["ms_per_step"] * 1e-3 + coded_q["h2d_ms"] * 1e-3 + t_down) / 1e9, 3)
                                                   if coded_q else None),
        "value_incl_pcie_coded_download": round(bases_local / (step_s + h2d_s + (t_down + t_dec) * len(batches)) / 1e9, 3),
        "value_incl_pcie_coded_copy_only": round(bases_local / (step_s + h2d_s + t_down * len(batches)) / 1e9, 3),
        "value_incl_pcie_plain_download": round(bases_local / (step_s + h2d_s + t_down_plain * len(batches)) / 1e9, 3),
        "note": "value = device-side expansion (P0) + every kernel of the run, raw arrays resident in HBM (a contig pays both, "
                "once); value_kernels_only = the run's kernels alone (sum of HIP-event stage times; rounds 1-3 reported this as "
                "`value`); value_incl_pcie adds the host->device copy (pageable numpy memory; bases as BAM packs them, "
                "crumble_batch.bases4, and BAM's per-read lengths instead of offset tables) and the CODED download "
                "(crumble_hip_encode + crumble_hip_download_coded: 2 bits per base + escape bytes, then crumble_decode_quals on "
                "the host's threads -- byte for byte the plain download, checked: coded_download_equals_plain); d2h_plain_ms "
                "is the one-byte-per-base download (steady state into mapped pageable buffers; d2h_first_ms also pays the page faults); "
                "value_incl_pcie takes whichever of the two is faster on this host for a FLAT output array (the coded form is "
                "made for callers that expand it into their records while they write them, crumble-hip's writer: "
                "value_incl_pcie_coded_copy_only leaves the expansion to that pass); coded_qualities: the same contig uploaded with its "
                "qualities as 2- / 4-bit codes (crumble_pack_quals + crumble_batch.qualsp, k_p0_quals inside the step): value_incl_pcie "
                "takes that way in when packing + copy + step beat the plain copy + step; value_incl_pcie_coded_both_copies_only = "
                "coded in and out with the host-side packing / expansion left to the reader's and writer's record passes",
    }

    out = {
        "metric": "Gbases/s quality-quantized, 30x WGS; bit-exact vs reference crumble",
        "value": round(value, 4), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "u8/f64", "data": "synthetic",
        "config": {"workload": (f"{label} synthetic contig ({cfg.contig_len} bp, "
                                f"{hb.n_reads} reads, {hb.n_bases} bases per GPU), "
                                f"{'-' + str(args.preset) if args.preset else 'default level'}"
                                f"{' -y ' + tech if tech else ''}"
                                f"{' -R keep.bed (' + str(len(hb.bed_start)) + ' regions)' if len(hb.bed_start) else ''}"
                                if args.workload != "wgs" else
                                f"{label}: 24 synthetic contigs (GRCh38 lengths x {args.scale}), {total_bases} bases "
                                f"LPT-packed on {world} GPU(s), -{args.preset}"),
                   "preset": args.preset, "tech": tech, "scale": args.scale,
                   "contigs": world if args.workload != "wgs" else 24, "stats_allreduce": stats_via,
                   **({"lpt_max_over_mean_load": round(lpt_imbalance, 4)} if args.workload == "wgs" else {}),
                   "parity": "every parity check is against oracle/ (a CPU restatement of crumble 0.9.1's transcode(); "
                             "reference-pinned for str_finder.c only: htslib is absent offline)",
                   "gen_s": round(t_gen, 1), "upload_s": round(t_up, 2)},
        "roofline": roofline,
        "around_the_step": extra,
    }

    if world > 1 and args.workload == "chr20" and not args.profile:
        # BASELINE config 5 is the configuration that names N GPUs: the driver's `--gpus N` line keeps the chr20 weak-scaling
        # headline and measures config 5 beside it (every rank takes part: collectives inside)
        for e_ in engines:
            e_.close()
        engines = []
        del tq, tf
        try:
            ws = measure_wgs_strong(ca, synth, shard, dist, rank, world, local_rank, red_dev, args.wgs_strong_scale * args.scale,
                                    preset=args.preset or 5)
        except Exception as ex:                    # noqa: BLE001
            ws = {"error": repr(ex)[:300]}
        out["wgs_strong"] = ws
    if rank == 0 and args.workload != "wgs":
        # compression proxy (no CRAM writer offline): share of the quality bytes the pass
        # rewrote and the order-0 entropy of the quality bytes before / after, from the engine's own statistics
        names = ca.capi.ST_NAMES
        h_out = stats[ca.capi.ST_HIST0:ca.capi.ST_HIST0 + 256].astype(np.float64) / max(1, args.steps * world)
        n_in = min(hb.n_bases, 200_000_000)
        h_in = np.bincount(hb.quals[:n_in], minlength=256).astype(np.float64)

        def entropy(h):
            p = h[h > 0] / h.sum()
            return float(-(p * np.log2(p)).sum()) if h.sum() else 0.0
        out["compression_proxy"] = {
            "fraction_rewritten": round(float(stats[names.index("bases_changed")]) / max(1.0, float(stats[names.index("bases")])), 5),
            "entropy_in_bits_per_quality": round(entropy(h_in), 4), "entropy_out_bits_per_quality": round(entropy(h_out), 4),
            "note": f"order-0 entropy; input histogram over the first {n_in} qualities of rank 0's contig, output from the "
                    "statistics vector (all ranks)"}
    if rank == 0:
        # ---- parity of the TIMED result + CPU baseline on a bounded prefix of the same workload ----
        import oracle_lib as O
        ns_per_base = 20e-9                      # (the restatement runs at ~0.05 Gbases/s on the GPU boxes' hosts)
        n_sample_bases = args.cpu_seconds / ns_per_base
        n_sample = int(min(hb.n_reads, max(1000, n_sample_bases / max(1, hb.n_bases / max(1, hb.n_reads)))))
        sample = prefix_batch(hb, n_sample)
        if len(hb.bed_start):
            sample.bed_start, sample.bed_end = hb.bed_start, hb.bed_end
        if not args.no_cpu and world == 1:          # CPU baseline: rank 0 at N=1 only
            O.load()
            t0 = time.perf_counter()
            wq, wf, ws = O.contig(params, sample)
            t_cpu = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": round(sample.n_bases / t_cpu / 1e9, 5), "unit": "Gbases/s", "cores": 1,
                "kind": "port",
                "sample": f"first {sample.n_reads} reads / {sample.n_bases} bases of the same contig, "
                          f"{t_cpu:.1f} s, CPU restatement (oracle/) single thread like reference crumble",
            }
            if not args.no_verify:
                # (a) the timed engine's own output: reads whose columns lie wholly inside the sample have their
                #     exact bytes in the oracle's run on the sample alone
                k = interior_reads(hb, n_sample)
                nbk = int(hb.seq_off[k])
                ok_q = bool(np.array_equal(tq[:nbk], wq[:nbk]))
                ok_f = bool(np.array_equal(tf[:k], wf[:k]))
                # (b) the same engine object, reused for another batch (the sample): everything incl. statistics
                gq, gf, gs = eng.contig(sample)
                out["config"]["verified_timed_output_leading_reads"] = k if (ok_q and ok_f) else False
                out["config"]["verified_bit_exact_on_sample"] = bool(
                    np.array_equal(gq, wq) and np.array_equal(gf, wf) and np.array_equal(gs, ws))
            if not args.no_cpu_all:
                import multiprocessing as mp
                cores = _host_threads()
                per = max(1000, n_sample // 4)
                jobs = [(window_batch(hb, (k * (hb.n_reads // max(1, cores))) % max(1, hb.n_reads - per), per),
                         args.preset, tech) for k in range(cores)]
                t0 = time.perf_counter()
                with mp.get_context("fork").Pool(cores) as pool:
                    res = pool.map(_cpu_worker, jobs)
                wall = time.perf_counter() - t0
                out["cpu_baseline_all"] = {
                    "value": round(sum(r[0] for r in res) / wall / 1e9, 5), "unit": "Gbases/s", "cores": cores,
                    "kind": "port",
                    "sample": f"{cores} independent windows of {per} reads of the same contig, one process per core "
                              f"(contigs / windows are the independent unit), {wall:.1f} s wall",
                }
        if world == 1 and not args.profile:
            try:
                out["small_contig"] = measure_small_contig(ca, synth, local_rank, preset=args.preset or 5)
            except Exception as ex:                # noqa: BLE001
                out["small_contig"] = {"error": repr(ex)[:300]}
        if world == 1 and (args.e2e or (not args.no_e2e and args.workload == "chr20")):
            # metric (ii), file -> file (never part of `value`)
            try:
                out["end_to_end"] = end_to_end_legs(ca, synth, shard, args, hb, label, local_rank)
            except Exception as ex:                # noqa: BLE001
                out["end_to_end"] = {"error": repr(ex)[:300]}
        if world == 1 and args.workload == "chr20" and args.scale == 1.0 and not args.no_other_configs and not args.no_cpu:
            # the other single-GPU configurations of BASELINE.json, measured by the same run (never part of `value`)
            for e_ in engines:
                e_.close()
            engines = []
            del tq, tf
            oc = {}
            for key, wl, sc in (("config3_panel", "panel", 1.0), ("config4_hifi", "hifi", 1.0)):
                try:
                    oc[key] = measure_config(ca, synth, wl, sc, local_rank)
                except Exception as ex:            # noqa: BLE001
                    oc[key] = {"error": repr(ex)[:300]}
            try:
                oc["config5_wgs_streamed"] = measure_wgs_streamed(ca, synth, shard, args.wgs_stream_scale, local_rank)
            except Exception as ex:                # noqa: BLE001
                oc["config5_wgs_streamed"] = {"error": repr(ex)[:300]}
            out["other_configs"] = oc
        print(json.dumps(out))
    for e_ in engines:
        e_.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
