export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import synth
from crumble_amd import bamio
bamio.write_batches("/tmp/tiny.bam", [("chr20", 64444167)], [synth.generate(synth.config("chr20", scale=0.0005))], level=1)
PY
for e in "CRUMBLE_HIP_ENGINES=1" "CRUMBLE_HIP_ENGINES=4" "CRUMBLE_HIP_PIPELINE=host CRUMBLE_HIP_ENGINES=1" "CRUMBLE_HIP_PIPELINE=host CRUMBLE_HIP_CODEC=host CRUMBLE_HIP_ENGINES=1"; do
  for rep in 1 2 3; do
    t0=$(date +%s%N); env $e crumble_amd/crumble-hip -5 -z /tmp/tiny.bam /tmp/tiny_out.bam; t1=$(date +%s%N)
    echo "$e: $(( (t1 - t0) / 1000000 )) ms"
  done
done
