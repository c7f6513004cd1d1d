#!/usr/bin/env python3
"""Turn gpurun_out/<tag>_{bench.json,prof,pmc_fetch,pmc_write} (written by tools/profile_round.sh on the
GPU box) into the tracked summaries profiles/<tag>_{bench_chr20_full.json,kernel_stats_chr20_full.csv,
pmc_traffic.json}.  Usage: python tools/summarize_profiles.py r01d"""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


line = [l for l in open(os.path.join(go, f"{tag}_bench.json")) if l.startswith("{")][-1]
open(os.path.join(pr, f"{tag}_bench_chr20_full.json"), "w").write(line)
ks = glob.glob(os.path.join(go, f"{tag}_prof", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(pr, f"{tag}_kernel_stats_chr20_full.csv"))
res = collections.defaultdict(dict)
for key, sub, ctr in (("FETCH_SIZE_KB_per_launch", "pmc_fetch", "FETCH_SIZE"), ("WRITE_SIZE_KB_per_launch", "pmc_write", "WRITE_SIZE")):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(os.path.join(go, f"{tag}_{sub}", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in tot:
        res[k][key] = tot[k] / n[k]
        res[k]["launches"] = n[k]
if res:
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), chr20 30x full scale, raw counter "
               "values in KB per launch; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide (16 B/lane) streaming "
               "reads by 2x, other widths uncalibrated", "kernels": res},
              open(os.path.join(pr, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(line.strip()[:400])
