set -u
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do for fill in 1 0; do
  SPR_GAP_FILLER=$fill python3 bench.py --workload c5s --steps 12 --warmup 4 --no-cpu 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('c5s fill=$fill', d['ms_per_step'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'})"
done; done
for fill in 1 0; do
  SPR_GAP_FILLER=$fill python3 bench.py --workload c5 --steps 6 --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('c5 fill=$fill', d['ms_per_step'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'})"
done
