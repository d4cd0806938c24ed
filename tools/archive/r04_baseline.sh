#!/bin/bash
# GPU box, round 4 first call: (1) device timeline of one rank's block of config 4 at N = 8 (what issues the fill / copy
# calls, what sits between the kernels), (2) SPR_TRACE host marks at that shape, (3) SQ / GRBM counters of the dominant
# kernels at config 3 and at the block (MFMA busy, issue stalls, LDS instructions, held clock), (4) plain bench lines.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04a
mkdir -p $out
SHARE="--workload c4 --share-of 8 --share-rank 3"
echo "== timeline c4 share"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tl_c4share -o run -- python3 bench.py $SHARE --steps 6 --warmup 2 --no-cpu > $out/tl_c4share.json 2> $out/tl_c4share.err || tail -3 $out/tl_c4share.err
python3 tools/timeline.py $out/tl_c4share > $out/tl_c4share.txt 2>&1; tail -25 $out/tl_c4share.txt
echo "== SPR_TRACE c4 share"
SPR_TRACE=1 python3 bench.py $SHARE --steps 6 --warmup 2 --no-cpu > $out/trace_c4share.json 2> $out/trace_c4share.err; grep "spr trace" $out/trace_c4share.err | tail -4
echo "== bench c4 share (plain, extra)"
python3 bench.py $SHARE --steps 20 --warmup 5 --no-cpu --extra > $out/bench_c4share.json 2> $out/bench_c4share.err; cat $out/bench_c4share.json | cut -c1-1500
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM")
for wl in c3 c4share; do
  args="--workload c3"; [ $wl = c4share ] && args="$SHARE"
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    echo "== pmc $wl set $i"
    rocprofv3 --pmc $set --output-format csv -d $out/pmc_${wl}_$i -o run -- python3 bench.py $args --steps 3 --warmup 1 --no-cpu > $out/pmc_${wl}_$i.json 2> $out/pmc_${wl}_$i.err || tail -3 $out/pmc_${wl}_$i.err
  done
  # kernel durations of the SAME kind of run (profiled passes clock differently from plain ones): kernel-trace stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o run -- python3 bench.py $args --steps 3 --warmup 1 --no-cpu > $out/stats_$wl.json 2> $out/stats_$wl.err || tail -3 $out/stats_$wl.err
  python3 tools/pmc_summary.py $(find $out/pmc_${wl}_* -name "*counter_collection.csv") > $out/pmc_$wl.txt 2>&1
  grep -E "stats_gram_own|project_ws|reconstruct" $out/pmc_$wl.txt | cut -c1-600
done
echo "== bench c3 (plain, extra)"
python3 bench.py --workload c3 --steps 10 --warmup 3 --extra > $out/bench_c3.json 2> $out/bench_c3.err; cat $out/bench_c3.json | cut -c1-2500
