#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of one bench configuration.  Usage: tools/kstats.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python3 $R/bench.py --steps 3 --warmup 1 --profile "$@" > $R/gpurun_out/${tag}_prof.log 2>&1
f=$(ls $R/gpurun_out/${tag}_prof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])/1e3:10.1f} us x{r['Calls']:>4}  {r['Percentage']:>6}%  {r['Name'][:70]}")
PY
