#!/bin/bash
# Runs ON THE GPU BOX: the file -> file legs of the default bench line alone (host codec, device codec, default level).
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout ${1:-400} python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-verify --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']
ks=('value','loop_s','read_parse_s','gpu_incl_pcie_s','deflate_write_s','out_bytes','codec','pipeline','gpu_stage_s')
for name,o in (('level 1 auto',e),('level 1 host pipeline',e.get('host_pipeline',{})),('level 1 host',e.get('host_codec',{})),('level 1 device',e.get('device_codec',{})),('default auto',e.get('default_level',{}).get('auto',{})),('default host',e.get('default_level',{}).get('host_codec',{})),('one contig',e.get('one_contig',{}))):
    print(name, {k:o.get(k) for k in ks})
print('small_contigs', json.dumps(e.get('small_contigs')))
print('codec_kernels', json.dumps(e.get('codec_kernels')))"
