#!/bin/bash
# Runs ON THE GPU BOX: short kernels-only bench lines for the main configurations.  QB_CONFIGS="chr20 panel hifi" selects.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in ${QB_CONFIGS:-chr20 panel hifi}; do
  case $cfg in
    chr20) args="--workload chr20 --preset 5";;
    chr20_9) args="--workload chr20 --preset 9";;
    panel) args="--workload panel --preset 9";;
    hifi)  args="--workload hifi --scale 0.1 --preset 9";;
  esac
  python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-verify --no-other-configs $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
done
