#!/bin/bash
# GPU box (round 5): the sharded path with the real kernels under 2-4 gloo ranks on one GPU -- now with the CU-free p2p field
# exchange (IPC peer-mapped buffers, SDMA pushes) --, then the bench rehearsals N = 2 / 4 that show the comm.gather_path keys.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_p2p}; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_dist_gpu_gloo.py -x -q > $out/tests_dist.log 2>&1; rc=$?; tail -15 $out/tests_dist.log
[ $rc -ne 0 ] && exit $rc
for n in 2 4; do
  SPR_BENCH_ONE_GPU=1 SPR_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus $n --workload c4s --steps 5 --warmup 2 > $out/rehearsal_n$n.json 2> $out/rehearsal_n$n.err || { tail -20 $out/rehearsal_n$n.err; exit 1; }
  python3 -c "
import json;d=json.load(open('$out/rehearsal_n$n.json'));print('rehearsal n=$n', d['ms_per_step'], d['comm']['gather_path'], {k:(v['ms_per_step'],v['ms_per_step_sync_gather'],v['gather_ms'],v['gather_exposed_ms']) for k,v in d['comm']['paths'].items()}, d.get('slowest_rank'))"
done
