# GPU box: cross kernel of the wide Gram path with constant-offset staging loads (default) against the clamped form (SPR_CROSS_VEC2=0),
# alternating in one call: c5s (f64, 16M rows x 512) and the config-5 share (f32, 100M rows x 512)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "wide or c5 or config5 or 512 or f32_storage" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do
  SPR_CROSS_VEC2=$v python3 bench.py --workload c5s --steps 8 --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vec2=$v c5s', d['ms_per_step'], [v_['ms'] for k,v_ in d['phases'].items() if k!='peaks'])"
done; done
for v in 0 1 0 1; do
  SPR_CROSS_VEC2=$v python3 bench.py --workload c5 --steps 4 --warmup 2 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vec2=$v c5', d['ms_per_step'], [v_['ms'] for k,v_ in d['phases'].items() if k!='peaks'])"
done
