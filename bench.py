#!/usr/bin/env python3
"""bench.py -- Gbases/s quality-quantized on a chr20-shaped 30x synthetic BAM (-5 preset).

One "step" = one full pass of the hot path (all kernels of crumble_hip_run) over
one contig batch that is already resident in HBM.  N>1: one process per GPU
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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


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


def file_to_file(hb, args, device):
    """Metric (ii): the same contig as a BAM file through the crumble-hip driver (BGZF inflate + BAM parse +
    H2D + kernels + D2H + BGZF deflate), wall clock of the whole process."""
    import re
    import subprocess
    import tempfile
    from crumble_amd import bamio
    tmp = tempfile.mkdtemp(prefix="crumble_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    src, dst = os.path.join(tmp, "in.bam"), os.path.join(tmp, "out.bam")
    try:
        bamio.write_batches(src, [("chr20", 64444167)], [hb], level=1)
        env = dict(os.environ, CRUMBLE_HIP_DEVICE=str(device), CRUMBLE_BAM_LEVEL="1")
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "crumble_amd", "crumble-hip"), "-v", f"-{args.preset}", src, dst],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if r.returncode:
            return {"error": r.stderr[-300:]}
        m = re.search(r"end-to-end ([0-9.]+) s \(read\+parse ([0-9.]+), upload\+kernels\+download ([0-9.]+), write ([0-9.]+)\)", r.stderr)
        st = [float(x) for x in m.groups()] if m else [None] * 4
        return {"value": round(hb.n_bases / wall / 1e9, 4), "unit": "Gbases/s", "process_wall_s": round(wall, 2),
                "loop_s": st[0], "read_parse_s": st[1], "gpu_incl_pcie_s": st[2], "deflate_write_s": st[3],
                "host_threads": _host_threads(), "host_cpus_visible": os.cpu_count(), "in_bytes": os.path.getsize(src), "out_bytes": os.path.getsize(dst),
                "note": "BAM (zlib level 1) -> crumble-hip -> BAM; includes process start, hipInit and engine creation"}
    finally:
        for f in (src, dst):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="chr20")
    ap.add_argument("--scale", type=float, default=1.0, help="contig length scale (1.0 = BASELINE config)")
    ap.add_argument("--preset", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--e2e", action="store_true",
                    help="also time metric (ii): BAM file -> crumble-hip -> BAM file (SURVEY.md 8d), reported "
                         "as an extra 'end_to_end' object; never part of 'value'")
    args = ap.parse_args()

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
    from crumble_amd import synth
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
        assign, _ = shard.lpt_pack(sizes, world)
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
        batches.append(synth.generate(cfg))
        scaling = "weak"
    t_gen = time.time() - t0
    hb = batches[0]
    params = ca.preset_params(args.preset)

    t0 = time.time()
    engines = []
    for b_ in batches:
        e_ = ca.Engine(params, local_rank)      # raises if no GPU (no CPU fallback)
        e_.upload(b_)                           # contigs stay resident in HBM
        engines.append(e_)
    eng = engines[0]
    t_up = time.time() - t0

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        for e_ in engines:
            e_.run()
    for e_ in engines:
        e_.stats_reset()
    sync_all()
    t0 = time.perf_counter()
    ktimes = {}
    for _ in range(args.steps):
        for e_ in engines:
            e_.run()                             # launches every kernel + stream sync
            for name, ms, by in e_.kernel_times():
                a = ktimes.setdefault(name, [0.0, 0.0, 0])
                a[0] += ms; a[1] += by; a[2] += 1
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the one collective of the path: statistics / quality histogram ----
    stats = sum(e_.stats_read() for e_ in engines)
    bases_local = sum(b_.n_bases for b_ in batches)
    if dist is not None:
        st = torch.from_numpy(stats.copy()).to(red_dev)
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
        stats = st.cpu().numpy()
        tb = torch.tensor([bases_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = int(tb.item())
    else:
        total_bases = bases_local
    assert int(stats[ca.capi.ST_NAMES.index("bases")]) == total_bases * args.steps, "stats all-reduce mismatch"

    value = total_bases * args.steps / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel (HIP events on the engine's stream) ----
    dom = max(ktimes.items(), key=lambda kv: kv[1][0])
    dom_ms = dom[1][0] / dom[1][2]
    dom_bytes = dom[1][1] / dom[1][2]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    algo_total = sum(3.0 * b_.n_bases + 32.0 * b_.n_reads + 8.0 * int(b_.cig_off[-1]) for b_ in batches)
    kernel_ms_total = sum(v[0] for v in ktimes.values()) / args.steps
    # HBM traffic of that kernel from rocprofv3 PMC counters (FETCH_SIZE + WRITE_SIZE, separate
    # passes, same command, committed under profiles/): per launch, bytes; None when not measured
    traffic = None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1:]:
            kk = json.load(open(f))["kernels"]
            key = [k for k in kk if k.split("<")[0] == dom[0]]
            if key and args.workload == "chr20" and args.scale == 1.0:
                t = kk[key[0]]
                traffic = round((t["FETCH_SIZE_KB_per_launch"] + t["WRITE_SIZE_KB_per_launch"]) * 1024)
    except Exception:
        traffic = None
    roofline = {
        "bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "kernel_ms": round(dom_ms, 3),
        "pipeline_achieved": round(algo_total / (kernel_ms_total * 1e-3) / 1e9, 2),
        "pipeline_frac": round(algo_total / (kernel_ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "kernels_ms": {k: round(v[0] / v[2], 3) for k, v in ktimes.items()},
    }

    out = {
        "metric": "Gbases/s quality-quantized, 30x WGS; bit-exact vs reference crumble",
        "value": round(value, 4), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "u8/f64", "data": "synthetic",
        "config": {"workload": (f"{args.workload}-30x synthetic contig ({cfg.contig_len} bp, "
                                f"{hb.n_reads} reads, {hb.n_bases} bases per GPU), preset -{args.preset}"
                                if args.workload != "wgs" else
                                f"wgs-30x: 24 synthetic contigs (GRCh38 lengths x {args.scale}), {total_bases} bases "
                                f"LPT-packed on {world} GPU(s), preset -{args.preset}"),
                   "preset": args.preset, "scale": args.scale,
                   "contigs": world if args.workload != "wgs" else 24,
                   "gen_s": round(t_gen, 1), "upload_s": round(t_up, 2)},
        "roofline": roofline,
    }

    if rank == 0:
        # ---- parity spot check + CPU baseline on a bounded prefix of the same workload ----
        import oracle_lib as O
        ns_per_base = 60e-9
        n_sample_bases = args.cpu_seconds / ns_per_base
        n_sample = int(min(hb.n_reads, max(1000, n_sample_bases / max(1, hb.n_bases / max(1, hb.n_reads)))))
        sample = prefix_batch(hb, n_sample)
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
                e2 = ca.Engine(params, local_rank)
                gq, gf, gs = e2.contig(sample)
                e2.close()
                out["config"]["verified_bit_exact_on_sample"] = bool(
                    np.array_equal(gq, wq) and np.array_equal(gf, wf) and np.array_equal(gs, ws))
        if args.e2e:
            out["end_to_end"] = file_to_file(hb, args, local_rank)
        print(json.dumps(out))
    for e_ in engines:
        e_.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
