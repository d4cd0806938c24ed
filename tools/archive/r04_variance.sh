#!/bin/bash
# GPU box: the two headline lines three times each in one call (within-box spread); run from several calls for the box-to-box spread
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do
  python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('c3     ', d['ms_per_step'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'}, d['placement_ms'])"
  python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('c4share', d['ms_per_step'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'}, d['placement_ms'])"
done
