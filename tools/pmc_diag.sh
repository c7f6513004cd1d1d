#!/bin/bash
# Runs ON THE GPU BOX: instruction-cache / scalar-cache / issue counters of one kernel (separate --pmc passes, no
# trace domains) for the default bench workload.  Usage: tools/pmc_diag.sh <tag> [kernel[,kernel...]] [bench args...]
# (kernel = prefix of the kernel's name; per-launch averages; also written to gpurun_out/<tag>_diag.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-diag}; kern=${2:-k_columns}; shift; shift
cd /tmp && export TMPDIR=/tmp
n=0
pass() { n=$((n+1)); rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_diag_$n -- python3 $R/bench.py --steps 1 --warmup 0 --profile $EXTRA > /dev/null 2>&1; }
EXTRA="$*"
pass SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN
pass SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
python3 - "$R/gpurun_out" "$tag" "$kern" <<'PY'
import csv, glob, json, sys, collections
root, tag, kerns = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
res = {}
for kern in kerns:
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"{root}/{tag}_diag_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].split("(")[0].replace("void ", "").startswith(kern):
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    res[kern] = {k: round(tot[k] / n[k]) for k in sorted(tot)}
    print("==", kern)
    for k in sorted(tot):
        print(f"{k:28s} {tot[k]/n[k]:16.0f}  (x{n[k]})")
json.dump(res, open(f"{root}/{tag}_diag.json", "w"), indent=1)
PY
