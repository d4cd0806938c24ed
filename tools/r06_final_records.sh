#!/bin/bash
# GPU box (round 6): the bench records kept under profiles/ for the final tree -- the driver's command (default flags), the
# per-workload lines with --extra, one rank's block of config 4 at N = 8 (default = deferred launches, and without), c3k, c5s, c5
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_final; mkdir -p $out
run() { name=$1; shift; timeout -k 10 600 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; }; python3 -c "
import json;d=json.load(open('$out/$name.json'));print('$name', d['ms_per_step'], d.get('ms_per_step_pipelined'), d['value'], d['hbm_roofline_frac_step'], {k:round(v['ms'],4) for k,v in d['phases'].items() if k!='peaks'}, d['gaps_ms'], d['placement_ms'], (d.get('cpu_baseline') or {}).get('value'), (d.get('parity') or {}).get('sensors_equal'), (d.get('parity') or {}).get('field_rel_fro'))"; }
run bench_default
run bench_c2_extra --workload c2 --steps 300 --warmup 30 --extra
run bench_c1_extra --workload c1 --steps 300 --warmup 30 --extra
run bench_c4_share8_rank3_p2p_loopback7 --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7
run bench_c4_share8_rank3_p2p_loopback7_nodefer --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --p2p-loopback 7 --no-defer-reconstruct
run bench_c4_share8_rank3_rccl_1rank --workload c4 --share-of 8 --share-rank 3 --steps 30 --warmup 5 --no-cpu --gather rccl
run bench_c3_extra --workload c3 --steps 20 --warmup 5 --extra
run bench_c3k --workload c3k --steps 10 --warmup 3 --no-cpu
run bench_c5s_extra --workload c5s --steps 10 --warmup 3 --extra
run bench_c5_extra --workload c5 --steps 3 --warmup 1 --extra
