#!/bin/bash
# GPU box (round 6): the p2p exchange after the poison / per-stream batching change -- sharded GPU tests, the nesting test of the
# downloads, and one rank's block of config 4 with the loopback exchange (host cost of a push: comm.p2p_host_ms_per_gather)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r06_p2p}; mkdir -p $out
timeout -k 10 700 python3 -m pytest tests/test_dist_gpu_gloo.py "tests/test_gpu_parity.py::test_downloads_nest" "tests/test_gpu_parity.py::test_small_downloads_by_kernel_and_ticket" "tests/test_gpu_parity.py::test_engine_close_and_reuse" -x -q > $out/tests.log 2>&1; rc=$?; tail -15 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
timeout -k 10 200 python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7 > $out/p2p_loop7_$rep.json 2> $out/p2p_loop7_$rep.err || { tail -5 $out/p2p_loop7_$rep.err; exit 1; }
python3 -c "
import json;d=json.load(open('$out/p2p_loop7_$rep.json'));print('loop7', d['ms_per_step'], d['ms_per_step_sync_gather'], d['comm'].get('p2p_host_ms_per_gather'), {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'}, d['gaps_ms'])"
done
