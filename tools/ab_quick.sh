#!/bin/bash
# Runs ON THE GPU BOX: A/B of libraries made by tools/build_variant.sh on one box, one short bench each.
# Usage: tools/ab_quick.sh "<tag> <tag> ..." [bench args...]   -> value, ms/step, kernel stages per tag
R=${GRAFT_REPO_ROOT:-$(pwd)}
tags=$1; shift
for tag in $tags; do
  echo "== $tag $*"
  CRUMBLE_HIP_LIB=$R/build/libcrumble_$tag.so timeout 150 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-verify --no-other-configs --no-e2e "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms'])"
done
