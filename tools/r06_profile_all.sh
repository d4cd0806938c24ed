#!/bin/bash
# GPU box, round 6 evidence: kernel stats + FETCH / WRITE / SQ counters per workload (tools/profile_round.sh), then the plain
# bench lines (filler on, --extra) that profiles/r06_bench_*.json record.  usage: tools/r06_profile_all.sh [workloads...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
WLS=${@:-"c1 c2 c3 c4share c5s c5"}
for w in $WLS; do
  case $w in
    c4share) spec="c4:--share-of 8 --share-rank 3"; args="--workload c4 --share-of 8 --share-rank 3";;
    *) spec="$w"; args="--workload $w";;
  esac
  bash tools/profile_round.sh r06 "$spec" > gpurun_out/r06_profile_$w.log 2>&1
  tail -4 gpurun_out/r06_profile_$w.log | cut -c1-200
  steps="--steps 20 --warmup 5"; [ $w = c5 ] && steps="--steps 5 --warmup 2"; [ $w = c1 ] && steps="--steps 300 --warmup 20"; [ $w = c2 ] && steps="--steps 200 --warmup 20"
  cpu=""; [ $w = c4share ] && cpu="--no-cpu"
  python3 bench.py $args $steps --extra $cpu > gpurun_out/r06_bench_$w.json 2> gpurun_out/r06_bench_$w.err || tail -3 gpurun_out/r06_bench_$w.err
  python3 -c "
import json;d=json.load(open('gpurun_out/r06_bench_$w.json'));print('$w', d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'}, d['placement_ms'])"
done
