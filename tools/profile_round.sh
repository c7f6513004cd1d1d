#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench JSON, rocprofv3 kernel stats, and the two PMC traffic passes
# for the default bench workload.  Usage: tools/profile_round.sh <tag>     (e.g. r01d)
# Outputs land in gpurun_out/<tag>_*; tools/summarize_profiles.py turns them into profiles/<tag>_*.
set -u
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# (the bench line first: a driver process that starts right behind other GPU processes' exits -- the calibration's passes here --
# can wait in its first allocations, profiles/NOTES_r06.md)
python3 $R/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
bash $R/tools/fetch_calib.sh $tag > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -- python3 $R/bench.py --steps 5 --warmup 1 --profile > $out/${tag}_prof.log 2>&1
# PMC passes on their own (no trace domains), one counter per pass
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --profile > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --profile > /dev/null 2>&1
tail -1 $out/${tag}_bench.json
