R=${GRAFT_REPO_ROOT:-$(pwd)}
for tag in one two; do
  echo "== $tag"
  CRUMBLE_HIP_LIB=$R/build/libcrumble_$tag.so python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-verify --no-other-configs --no-e2e 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms']['k_columns2'])"
done
