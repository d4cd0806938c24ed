#!/bin/bash
# GPU box (round 5): one rank's block of config 4 at N = 8, (a) as in round 4 (1-rank RCCL group), (b) with the p2p exchange and 7
# imaginary peers inside this GPU: the 7 x 90 MB SDMA pushes of a real 8-rank exchange in flight under the next Gram pass.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_c4share}; mkdir -p $out
for rep in 1 2; do
python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --gather rccl > $out/rccl_$rep.json 2> $out/rccl_$rep.err || tail -5 $out/rccl_$rep.err
python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7 > $out/p2p_loop7_$rep.json 2> $out/p2p_loop7_$rep.err || tail -5 $out/p2p_loop7_$rep.err
done
python3 - <<PY
import json
for name in ('rccl_1', 'p2p_loop7_1', 'rccl_2', 'p2p_loop7_2'):
    d = json.load(open('$out/%s.json' % name))
    print(name, 'step', d['ms_per_step'], 'sync', d['ms_per_step_sync_gather'], {k: v['ms'] for k, v in d['phases'].items() if k != 'peaks'},
          'comm', {k: d['comm'][k] for k in ('allreduce_ms', 'gather_ms', 'gather_exposed_ms', 'gather_path')})
PY
