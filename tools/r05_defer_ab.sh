#!/bin/bash
# GPU box (round 5): bench.py with and without --defer-reconstruct, alternating in one call
set -u
cd "${GRAFT_REPO_ROOT:-.}"
run() { python3 bench.py "$@" --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*'.replace('--workload ','').replace('--steps','K').replace('--warmup','W'), '|', d['ms_per_step'], d['gaps_ms'], [round(v['ms'],4) for k,v in d['phases'].items() if k!='peaks'], d['comm'].get('gather_exposed_ms'))"; }
for rep in 1 2; do
  run --workload c2 --steps 300 --warmup 30
  run --workload c2 --steps 300 --warmup 30 --defer-reconstruct
  run --workload c1 --steps 300 --warmup 30
  run --workload c1 --steps 300 --warmup 30 --defer-reconstruct
  run --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --p2p-loopback 7
  run --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --p2p-loopback 7 --defer-reconstruct
done
run --workload c3 --steps 10 --warmup 3
run --workload c3 --steps 10 --warmup 3 --defer-reconstruct
