#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_{bench.json,prof,pmc_fetch,pmc_write} (written by tools/profile_round.sh on the
GPU box) into the tracked summaries profiles/<tag>_{bench_chr20_full.json,kernel_stats_chr20_full.csv,
pmc_traffic.json}.  Usage: python tools/summarize_profiles.py r01d"""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_source_sha          # the kernels these counters were measured on (bench.py looks for a match)
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


line = [l for l in open(os.path.join(go, f"{tag}_bench.json")) if l.startswith("{")][-1]
open(os.path.join(pr, f"{tag}_bench_chr20_full.json"), "w").write(line)
ks = glob.glob(os.path.join(go, f"{tag}_prof", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(pr, f"{tag}_kernel_stats_chr20_full.csv"))
def traffic(suffix):
    """FETCH_SIZE / WRITE_SIZE per launch and kernel from the two --pmc passes gpurun_out/<tag>_pmc_{fetch,write}<suffix>"""
    res = collections.defaultdict(dict)
    for key, sub, ctr in (("FETCH_SIZE_KB_per_launch", "pmc_fetch", "FETCH_SIZE"), ("WRITE_SIZE_KB_per_launch", "pmc_write", "WRITE_SIZE")):
        tot, n = collections.defaultdict(float), collections.defaultdict(int)
        for f in glob.glob(os.path.join(go, f"{tag}_{sub}{suffix}", "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    k = short(r["Kernel_Name"])
                    tot[k] += float(r["Counter_Value"]); n[k] += 1
        for k in tot:
            res[k][key] = tot[k] / n[k]
            res[k]["launches"] = n[k]
    return res


res = traffic("")
other = {name: {k: v for k, v in traffic("_" + name).items() if k.startswith("k_columns2") or k.startswith("k_replay_fast")}
         for name in ("panel", "hifi")}
other = {k: v for k, v in other.items() if v}
if res:
    calib = None
    cf = os.path.join(go, f"{tag}_fetch_calib.json")
    if os.path.exists(cf):
        calib = json.load(open(cf))
        shutil.copy(cf, os.path.join(pr, f"{tag}_fetch_calib.json"))
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), chr20 30x full scale, raw counter "
               "values in KB per launch.  FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies a 128-byte request as 64 "
               "(MI355X_MICROARCH.md); tools/fetch_calib.* measured counter / bytes = 0.50 for 16-B streams and for "
               "32-byte records, 0.70 for the column kernel's 20-byte segment loads (which re-touch half of every line "
               "one tile later): bytes moved = 2 x FETCH_SIZE; WRITE_SIZE is exact",
               "source_sha16": kernel_source_sha(), "fetch_calibration": calib["kernels"] if calib else None, "kernels": res,
               "other_configs": other},
              open(os.path.join(pr, f"{tag}_pmc_traffic.json"), "w"), indent=1)
# the other configurations: bench line + rocprofv3 kernel stats each
for name in ("chr20_9", "panel", "hifi_config4"):
    bj = os.path.join(go, f"{tag}_bench_{name}.json")
    if os.path.exists(bj):
        ls = [l for l in open(bj) if l.startswith("{")]
        if ls:
            open(os.path.join(pr, f"{tag}_bench_{name}.json"), "w").write(ls[-1])
    ks2 = glob.glob(os.path.join(go, f"{tag}_prof_{name}", "*", "*_kernel_stats.csv"))
    if ks2:
        shutil.copy(ks2[0], os.path.join(pr, f"{tag}_kernel_stats_{name}.csv"))
# the device BGZF codec on a gigabyte of records: the probe's line and the kernel stats of the same command
cj = os.path.join(go, f"{tag}_codec_probe.json")
if os.path.exists(cj):
    ls = [l for l in open(cj) if l.startswith("{")]
    if ls:
        open(os.path.join(pr, f"{tag}_codec_probe.json"), "w").write(ls[-1])
ks3 = glob.glob(os.path.join(go, f"{tag}_prof_codec", "*", "*_kernel_stats.csv"))
if ks3:
    shutil.copy(ks3[0], os.path.join(pr, f"{tag}_kernel_stats_codec.csv"))
# the device pipeline: the driver's own lines, kernel stats and timeline of one traced run, the inflate sweep
for name in ("pipeline.txt", "pipeline_timeline.txt", "inflate_sweep.txt"):
    f = os.path.join(go, f"{tag}_{name}")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(pr, f"{tag}_{name}"))
ks4 = glob.glob(os.path.join(go, f"{tag}_prof_pipeline", "*", "*_kernel_stats.csv"))
if ks4:
    shutil.copy(ks4[0], os.path.join(pr, f"{tag}_kernel_stats_pipeline.csv"))
# SQ counters of the column kernel, per launch: chr20 -5, the 300x panel, config 4
sqall = {}
for name in ("chr20", "panel", "hifi"):
    sq = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(go, f"{tag}_sq_{name}_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]).startswith("k_columns2"):
                a = sq[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    if sq:
        sqall["k_columns2" if name == "chr20" else "k_columns2_" + name] = {k: round(v[0] / v[1]) for k, v in sorted(sq.items())}
if sqall:
    sqall["note"] = ("rocprofv3 --pmc, per launch (wave-level instruction counts; *_CYCLES summed over SEs): k_columns2 = chr20 30x -5, "
                     "k_columns2_panel = 300x panel -9, k_columns2_hifi = config 4 (HiFi 20x, -y pbccs -R)")
    sqall["source_sha16"] = kernel_source_sha()
    json.dump(sqall, open(os.path.join(pr, f"{tag}_sq_counters.json"), "w"), indent=1)
print(line.strip()[:400])
