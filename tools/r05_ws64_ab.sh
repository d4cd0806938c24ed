#!/bin/bash
# GPU box (round 5): config 2 (m = 64) with the general projection kernel (SPR_PROJECT_WS=0) | the W-stationary kernel at 1 / 2 / 3
# workgroups per CU (SPR_WS_WG_PER_CU), alternating in one call; the projection tests first
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_ws64_ab}; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "project or random_shapes or norms or fixture or golden" > $out/tests.log 2>&1; rc=$?; tail -4 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for v in "SPR_PROJECT_WS=0" "SPR_WS_WG_PER_CU=1" "SPR_WS_WG_PER_CU=2" "SPR_WS_WG_PER_CU=3" "SPR_WS_WG_PER_CU=4"; do
  env $v timeout -k 10 300 python3 bench.py --workload c2 --steps 300 --warmup 30 --no-cpu 2>$out/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], {k:round(v['ms'],4) for k,v in d['phases'].items() if k!='peaks'}, d['gaps_ms'], 'placement', d['placement_ms'], d['path']['placement_from_row_norms'], 'crc', d['path'].get('sensors_crc32'))" || { tail -5 $out/err.log; exit 1; }
done; done 2>&1 | tee $out/ab.txt
