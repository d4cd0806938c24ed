#!/bin/bash
# GPU box (round 6): sharded GPU tests after the trial / defer-default / poison changes, the two full-size oracle comparisons,
# then the N = 8 rank block of config 4 with the loopback exchange (default = deferred launches) and the default line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r06_gpu1}; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_dist_gpu_gloo.py "tests/test_gpu_parity.py::test_config2_full_size_vs_oracle" "tests/test_gpu_parity.py::test_config3_sample_vs_oracle" "tests/test_gpu_parity.py::test_deferred_reconstruct_on_the_device" "tests/test_gpu_parity.py::test_golden_fixture" tests/test_c_linkage.py -x -q -s > $out/tests.log 2>&1; rc=$?; tail -15 $out/tests.log; grep "HIP vs oracle" $out/tests.log
[ $rc -ne 0 ] && exit $rc
run() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return; }; python3 -c "
import json;d=json.load(open('$out/$name.json'));print('$name', d['ms_per_step'], d.get('ms_per_step_pipelined'), d['ms_per_step_sync_gather'], d['value'], d['hbm_roofline_frac_step'], {k:round(v['ms'],4) for k,v in d['phases'].items() if k!='peaks'}, d['gaps_ms'], d['placement_ms'], (d['comm'] or {}).get('p2p_host_ms_per_gather'))"; }
run c4_share8_rank3_p2p_loopback7 --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7
run c4_share8_rank3_p2p_loopback7_nodefer --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7 --no-defer-reconstruct
run c4_share8_rank3_p2p_loopback7_b --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7
run c2 --workload c2 --steps 300 --warmup 30
run c3_default
