#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes (no trace domains) for the default bench workload; prints the
# per-launch averages of the dominant kernel.  Usage: tools/pmc_sq.sh <tag> [kernel-substring]
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-sq}; kern=${2:-k_columns}
cd /tmp && export TMPDIR=/tmp
pass() { rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_pmc_$1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1; }
pass SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM
python3 - "$R/gpurun_out" "$tag" "$kern" <<'PY'
import csv, glob, sys, collections
root, tag, kern = sys.argv[1:4]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(f"{root}/{tag}_pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:24s} {tot[k]/n[k]:16.0f}  (x{n[k]})")
PY
