#!/bin/bash
# GPU box: kernel-trace stats + HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes) for a list
# of bench workloads.   usage: tools/profile_round.sh <round-tag> <workload>[:extra bench args] ...
# Output under gpurun_out/<round-tag>_<workload>/ : stats CSVs, counter CSVs, the bench JSON lines and pmc_summary.txt
set -u
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
  python3 tools/pmc_summary.py $(find "$out" -name "*counter_collection.csv") > "$out/pmc_summary.txt" 2>&1
  grep -E "stats_gram|project|reconstruct|gram_cross|rowstats" "$out/pmc_summary.txt" | cut -c1-200
  f=$(find "$out/stats" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -8 "$f" | cut -c1-220
done
