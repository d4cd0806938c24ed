#!/bin/bash
# GPU box (round 5): the bench records kept under profiles/ for the final tree -- default lines with --extra, and the opt-in variants
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05_final; mkdir -p $out
run() { name=$1; shift; python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; }; python3 -c "
import json;d=json.load(open('$out/$name.json'));print('$name', d['ms_per_step'], d['value'], d['hbm_roofline_frac_step'], {k:round(v['ms'],4) for k,v in d['phases'].items() if k!='peaks'}, d['gaps_ms'], d['placement_ms'])"; }
run bench_default
run bench_c2_extra --workload c2 --steps 300 --warmup 30 --extra
run bench_c2_defer --workload c2 --steps 300 --warmup 30 --defer-reconstruct --no-cpu
run bench_c1_extra --workload c1 --steps 300 --warmup 30 --extra
run bench_c1_defer --workload c1 --steps 300 --warmup 30 --defer-reconstruct --no-cpu
run bench_c4_share8_rank3_rccl_1rank --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --gather rccl
run bench_c4_share8_rank3_p2p_loopback7 --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7
run bench_c4_share8_rank3_p2p_loopback7_defer --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7 --defer-reconstruct
run bench_c3_extra --workload c3 --steps 20 --warmup 5 --extra
run bench_c3_defer --workload c3 --steps 20 --warmup 5 --defer-reconstruct --no-cpu
run bench_c3k --workload c3k --steps 10 --warmup 3 --no-cpu
