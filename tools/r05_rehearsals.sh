#!/bin/bash
# GPU box (round 5): rehearsals of the multi-rank bench, all ranks on ONE GPU over gloo (exercises code, measures nothing): N = 2, 3, 4
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_rehearsals}; mkdir -p $out
for n in 2 3 4; do
  extra=""; [ $n = 3 ] && extra="--headline-gather other"       # N = 3 also exercises the branch that reports the all-gather
  SPR_BENCH_ONE_GPU=1 SPR_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus $n --workload c4s --steps 5 --warmup 2 $extra > $out/rehearsal_n$n.json 2> $out/rehearsal_n$n.err || { tail -20 $out/rehearsal_n$n.err; exit 1; }
  python3 -c "
import json;d=json.load(open('$out/rehearsal_n$n.json'));print('rehearsal n=$n', d['ms_per_step'], '|', d['comm']['gather_path'], '|', {k:(v['ms_per_step'],v['ms_per_step_sync_gather']) for k,v in d['comm']['paths'].items()}, d['comm'].get('p2p_memory'))"
done
