#!/bin/bash
# GPU box (round 5): host-transfer tests, then bench.py --extra at configs 1 / 2 / c5s: reconstruct() into a host array (the reference's contract)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_tohost}; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "pinned or host or big_attributes or reconstruct or stage or golden or fixture or wide" > $out/tests.log 2>&1; rc=$?; tail -4 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for wl in c2 c1 c5s; do
  timeout -k 10 400 python3 bench.py --workload $wl --steps 50 --warmup 10 --no-cpu --extra 2>$out/err.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['extra']; print('$wl', d['ms_per_step'], d['gaps_ms'], {k:e[k] for k in e if 'host' in k})" || { tail -5 $out/err.log; exit 1; }
done 2>&1 | tee $out/ab.txt
