#!/bin/bash
# GPU box (round 6): the copy streams ordered behind the reconstruct kernel by a counter (default) or by an event (SPR_P2P_ORDER=event):
# sharded GPU tests with the default, the loopback stress (every imaginary peer's copy compared bit for bit), then the alternating A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r06_ready_ab}; mkdir -p $out
timeout -k 10 700 python3 -m pytest tests/test_dist_gpu_gloo.py -x -q > $out/tests.log 2>&1; rc=$?; tail -4 $out/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 tools/p2p_stress.py 400 7 > $out/p2p_stress.txt 2>&1 || { tail -5 $out/p2p_stress.txt; exit 1; }
grep defer_reconstruct $out/p2p_stress.txt
for rep in 1 2 3; do for ord in counter event; do
SPR_P2P_ORDER=$ord timeout -k 10 200 python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7 > $out/${ord}_$rep.json 2> $out/${ord}_$rep.err || { tail -5 $out/${ord}_$rep.err; exit 1; }
python3 -c "
import json;d=json.load(open('$out/${ord}_$rep.json'));print('$ord $rep', d['ms_per_step'], d['ms_per_step_sync_gather'], d['comm'].get('p2p_host_ms_per_gather'), {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'}, d['gaps_ms'])"
done; done | tee $out/ab.txt
