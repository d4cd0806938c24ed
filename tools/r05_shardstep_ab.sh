#!/bin/bash
# GPU box (round 5): the step of a SHARDED placement as two launches (orthogonalise; down-date + search fused) | three
# (SPR_QR_FUSED_STEPS=0), on one rank's block of config 4 and of config 5, alternating in one call; the placement and sharded tests first
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_shardstep_ab}; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "qr or placement or pivot or sensor or gem or golden or fixture or dist or shard" > $out/tests.log 2>&1; rc=$?; tail -4 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for wlargs in "--workload c4 --share-of 8 --share-rank 3" "--workload c5 --share-of 8 --share-rank 3"; do
for rep in 1 2; do for fused in 0 1; do
  SPR_QR_FUSED_STEPS=$fused timeout -k 10 300 python3 bench.py $wlargs --steps 3 --warmup 1 --no-cpu 2>$out/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused=$fused', d['config']['workload'][:14], 'placement_ms', d['placement_ms'], 'sweeps', d['pivot_sweeps'], d['path']['pivot_pool_sweeps'], 'min gap', '%.3e' % d['min_pivot_gap'], 'crc', d['path'].get('sensors_crc32'))" || { tail -5 $out/err.log; exit 1; }
done; done; done 2>&1 | tee $out/ab.txt
