#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel statistics of one bench configuration.  Usage: tools/kstat.sh <rows> <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
rows=${1:-25}; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstat_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat_prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-verify --no-other-configs "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
python3 - $rows <<EOF
import csv,glob,sys
f=glob.glob("/tmp/kstat_prof/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.reader(open(f)))[1:1+int(sys.argv[1])]: print("%-60s %5s %10.1f us" % (r[0][:60], r[1], float(r[3])/1e3))
EOF
