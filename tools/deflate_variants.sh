#!/bin/bash
# Runs ON THE GPU BOX: kernel time of the device deflate built without one of its parts (tools/build_variant.sh <tag> -DGZ_EXP_<PART>:
# NOA pass A, NOCMP the candidate compares, NOPARSE pass B, NOHUFF the two trees, NOENC the token coding) -- what a part costs is what
# the kernel gains without it (the streams are valid or not; only the time counts).  VARIANTS="noa nocmp ..." picks the builds.
export TMPDIR=/tmp
export GZ_PROBE_NOCHECK=1
for v in "" ${VARIANTS:-noa nocmp}; do
  if [ -n "$v" ]; then export CRUMBLE_HIP_LIB=build/libcrumble_$v.so; fi
  echo "variant [$v]"; timeout 200 python3 tools/gz_probe.py 0.2 1 2>&1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(v['kernels_ms'],v.get('comp_bytes')) for k,v in d.items() if k.startswith('deflate')})"
done
