set -u
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3; do for fill in 1 0; do
  SPR_GAP_FILLER=$fill python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('fill=$fill', d['ms_per_step'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'})"
done; done
