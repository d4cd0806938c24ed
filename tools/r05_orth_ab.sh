#!/bin/bash
# GPU box (round 5): the placement tests, then optimal_placement at the bench workloads with the Gram-Schmidt passes of a step as
# register tiles (SPR_QR_ORTH_TILE=1, the default) | chains of loads (=0), alternating in one call
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_orth_ab}; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "qr or placement or pivot or sensor or gem or golden or fixture" > $out/tests.log 2>&1; rc=$?; tail -4 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for wlargs in "--workload c3" "--workload c4 --share-of 8 --share-rank 3" "--workload c2" "--workload c1" "--workload c5 --share-of 8 --share-rank 3"; do
for rep in 1 2; do for tile in 0 1; do
  SPR_QR_ORTH_TILE=$tile timeout -k 10 300 python3 bench.py $wlargs --steps 3 --warmup 1 --no-cpu 2>$out/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tile=$tile', d['config']['workload'][:14], 'placement_ms', d['placement_ms'], 'sweeps', d['pivot_sweeps'], d['path']['pivot_pool_sweeps'], 'min gap', '%.3e' % d['min_pivot_gap'], 'crc', d['path'].get('sensors_crc32'))" || { tail -5 $out/err.log; exit 1; }
done; done; done 2>&1 | tee $out/ab.txt
