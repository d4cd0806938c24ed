set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04g; mkdir -p $out
SHARE="--workload c4 --share-of 8 --share-rank 3"
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "gap_filler or native_library or golden_fixture or end_to_end_vs_oracle or conditioning" > $out/tests.log 2>&1; tail -3 $out/tests.log
for fill in 1 0; do
  SPR_GAP_FILLER=$fill python3 bench.py $SHARE --steps 20 --warmup 5 --no-cpu > $out/bench_c4share_fill$fill.json 2> $out/bench_c4share_fill$fill.err || tail -5 $out/bench_c4share_fill$fill.err
  python3 -c "
import json;d=json.load(open('$out/bench_c4share_fill$fill.json'));print('fill=$fill', d['ms_per_step'], d['ms_per_step_sync_gather'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'}, d['comm'], d['placement_ms'], d['train_ms'], d['predict_ms'])"
done
SPR_TRACE=1 python3 bench.py $SHARE --steps 4 --warmup 2 --no-cpu > $out/trace.json 2> $out/trace.err; grep "spr trace" $out/trace.err | tail -3
for fill in 1 0; do
  SPR_GAP_FILLER=$fill python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu > $out/bench_c3_fill$fill.json 2> $out/bench_c3_fill$fill.err || tail -5 $out/bench_c3_fill$fill.err
  python3 -c "
import json;d=json.load(open('$out/bench_c3_fill$fill.json'));print('c3 fill=$fill', d['ms_per_step'], {k:v['ms'] for k,v in d['phases'].items() if k!='peaks'}, d['placement_ms'])"
done
