export TMPDIR=/tmp
for v in "" noa nocmp; do
  if [ -n "$v" ]; then export CRUMBLE_HIP_LIB=build/libcrumble_$v.so; fi
  echo "variant [$v]"; timeout 200 python3 tools/gz_probe.py 0.2 1 2>&1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(v['kernels_ms'],v.get('comp_bytes')) for k,v in d.items() if k.startswith('deflate')})"
done
