#!/bin/bash
# GPU box: the whole -m gpu suite in ONE process, then the two headline bench lines
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r04_tests}; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -4 $out/tests.log
[ $rc -ne 0 ] && exit $rc
python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 20 --warmup 5 --no-cpu > $out/bench_c4share.json 2> $out/bench_c4share.err || tail -5 $out/bench_c4share.err
python3 -c "
import json;d=json.load(open('$out/bench_c4share.json'));print('c4share', d['ms_per_step'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'}, d['placement_ms'])"
