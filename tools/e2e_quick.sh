#!/bin/bash
# Runs ON THE GPU BOX: the file -> file legs of the default bench line alone (device codec and host codec side by side).
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout ${1:-400} python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-verify --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']; print(json.dumps(e, indent=1))"
