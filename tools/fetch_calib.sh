#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE of the calibration kernels (tools/fetch_calib.hip) against their known byte counts;
# writes gpurun_out/<tag>_fetch_calib.json.  Usage: tools/fetch_calib.sh <tag> [GiB]
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-rXX}; gib=${2:-2}
cd /tmp && export TMPDIR=/tmp
$R/build/fetch_calib $gib > $R/gpurun_out/${tag}_fetch_calib.txt
rm -rf $R/gpurun_out/${tag}_calib_pmc
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_calib_pmc -- $R/build/fetch_calib $gib > /dev/null 2>&1
python3 - "$R/gpurun_out" "$tag" <<'PY'
import csv, glob, json, sys, collections
root, tag = sys.argv[1:3]
known = {}
for l in open(f"{root}/{tag}_fetch_calib.txt"):
    f = l.split()
    known[f[0]] = {"bytes": int(f[2]), "ms": float(f[f.index("ms") + 1])}
    if "requested" in f:
        known[f[0]]["requested_bytes"] = int(f[f.index("requested") + 1])
tot = collections.defaultdict(list)
for fn in glob.glob(f"{root}/{tag}_calib_pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] == "FETCH_SIZE":
            tot[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
out = {}
for k, v in tot.items():
    name = k.split("<")[0]
    if name in known:
        kb = v[-1]                                   # the second (warm code, cold data) launch
        out[name] = dict(known[name], FETCH_SIZE_KB=kb, counter_over_bytes=round(kb * 1024 / known[name]["bytes"], 4))
json.dump({"what": "rocprofv3 FETCH_SIZE (KB) of kernels that read a buffer of known size once (tools/fetch_calib.hip)", "kernels": out},
          open(f"{root}/{tag}_fetch_calib.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
