#!/bin/bash
# Runs ON THE GPU BOX: short kernels-only bench lines for the main configurations.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for args in "--workload chr20 --preset 5" "--workload panel --preset 9" "--workload hifi --scale 0.1 --preset 9"; do
  python3 $R/bench.py --steps 5 --warmup 1 --no-cpu $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
done
