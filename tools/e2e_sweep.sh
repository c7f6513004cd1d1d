#!/bin/bash
# Runs ON THE GPU BOX: file -> file timing of crumble-hip on the chr20-shaped 30x BAM for several
# host thread counts; prints the driver's own phase timers.  Usage: tools/e2e_sweep.sh [scale] [threads...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
scale=${1:-1.0}; shift
threads=${@:-"16 64 256"}
tmp=$(mktemp -d /tmp/e2e.XXXX)
python3 - <<PY
import sys; sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from crumble_amd import bamio
import synth
hb = synth.generate(synth.config("chr20", scale=$scale))
bamio.write_batches("$tmp/in.bam", [("chr20", 64444167)], [hb], level=1)
print("bases", hb.n_bases)
PY
for t in $threads; do
  echo "== threads $t"
  CRUMBLE_BAM_LEVEL=1 $R/crumble_amd/crumble-hip -v -5 -@ $t $tmp/in.bam $tmp/out.bam 2>&1 | tail -2
done
rm -rf $tmp
