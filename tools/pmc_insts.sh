#!/bin/bash
# Runs ON THE GPU BOX: instruction counts of k_columns2 for several library variants (build/libcrumble_<tag>.so from
# tools/build_variant.sh; "default" = the in-tree library).  Usage: tools/pmc_insts.sh <tag> [<tag>...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = default ]; then unset CRUMBLE_HIP_LIB; else export CRUMBLE_HIP_LIB=$R/build/libcrumble_$tag.so; fi
  rm -rf $R/gpurun_out/insts_$tag
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
      --output-format csv -d $R/gpurun_out/insts_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --profile > /dev/null 2>&1
  python3 - "$R/gpurun_out/insts_$tag" "$tag" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_columns2" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[2], " ".join(f"{k.replace('SQ_','')}={v/1e6:.1f}M" for k, v in sorted(tot.items())))
PY
done
