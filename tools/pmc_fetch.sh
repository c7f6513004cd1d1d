#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE (separate passes) of k_columns2 for library variants (build/libcrumble_<tag>.so
# from tools/build_variant.sh; "default" = the in-tree library).  Usage: tools/pmc_fetch.sh <tag> [<tag>...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = default ]; then unset CRUMBLE_HIP_LIB; else export CRUMBLE_HIP_LIB=$R/build/libcrumble_$tag.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/fetch_${tag}_$c
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/fetch_${tag}_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-verify --no-other-configs ${PMC_BENCH_ARGS:-} > /dev/null 2>&1
  done
  python3 - "$R/gpurun_out" "$tag" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(f"{sys.argv[1]}/fetch_{sys.argv[2]}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0].split("<")[0], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); n[k] += 1
for (k, c), v in sorted(tot.items()):
    if v / n[(k, c)] > 20000:
        print(f"{sys.argv[2]:10s} {k:22s} {c:11s} {v / n[(k, c)] / 1e6:9.3f} GB raw (x{n[(k, c)]})")
PY
done
