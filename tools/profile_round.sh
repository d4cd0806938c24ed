#!/bin/bash
# GPU box: kernel-trace stats + HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes) for a list
# of bench workloads.   usage: tools/profile_round.sh <round-tag> <workload>[:extra bench args] ...
# Output under gpurun_out/<round-tag>_<workload>/ : stats CSVs, counter CSVs, the bench JSON lines and pmc_summary.txt
# Round 4: a third counter pass (SQ / GRBM: MFMA-busy cycles, issue stalls, LDS instructions, held clock) and the gap filler of
# fit() switched off in every profiled pass (it launches the Gram kernel a second time on part of the rows, which would mix
# two launch sizes into the per-kernel means); the plain bench lines under profiles/ are taken with it on.
set -u
export SPR_GAP_FILLER=0
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; shift
for spec in "$@"; do
  wl=${spec%%:*}
  extra=""
  [ "$spec" != "$wl" ] && extra=${spec#*:}
  name=$(echo "$wl$extra" | tr -c 'A-Za-z0-9\n' '_')
  out=gpurun_out/${tag}_${name}
  mkdir -p "$out"
  echo "== $wl $extra -> $out"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 bench.py --workload $wl $extra --steps 3 --warmup 1 --no-cpu > "$out/bench_under_rocprof.json" 2> "$out/stats.err" || tail -3 "$out/stats.err"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d "$out/$ctr" -o run -- python3 bench.py --workload $wl $extra --steps 2 --warmup 1 --no-cpu > "$out/$ctr.json" 2> "$out/$ctr.err" || tail -3 "$out/$ctr.err"
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$out/SQ1" -o run -- python3 bench.py --workload $wl $extra --steps 2 --warmup 1 --no-cpu > "$out/SQ1.json" 2> "$out/SQ1.err" || tail -3 "$out/SQ1.err"
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d "$out/SQ2" -o run -- python3 bench.py --workload $wl $extra --steps 2 --warmup 1 --no-cpu > "$out/SQ2.json" 2> "$out/SQ2.err" || tail -3 "$out/SQ2.err"
  python3 tools/pmc_summary.py $(find "$out" -name "*counter_collection.csv") > "$out/pmc_summary.txt" 2>&1
  grep -E "stats_gram|project|reconstruct|gram_cross|rowstats" "$out/pmc_summary.txt" | cut -c1-200
  f=$(find "$out/stats" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -8 "$f" | cut -c1-220
done
