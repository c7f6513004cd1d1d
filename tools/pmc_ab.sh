#!/bin/bash
# GPU box: SQ counters of k_columns on the panel workload for two libraries (A/B).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in build/libcrumble_cur.so crumble_amd/libcrumble_hip.so; do
  tag=$(basename $lib .so)
  export CRUMBLE_HIP_LIB=$R/$lib
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/ab_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --workload panel --preset 9 > /dev/null 2>&1
  python3 - "$R/gpurun_out/ab_$tag" "$tag" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_columns" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[2], {k: int(v) for k, v in sorted(tot.items())})
PY
done
