#!/bin/bash
# CPU box: copy the summaries of a tools/r04_profile_all.sh run from gpurun_out/ into profiles/ (tracked).
# usage: tools/collect_profiles.sh r04
set -u
cd "$(dirname "$0")/.."
tag=${1:-r04}
for pair in "c1:${tag}_c1" "c2:${tag}_c2" "c3:${tag}_c3" "c4share8:${tag}_c4__share_of_8___share_rank_3" "c5s:${tag}_c5s" "c5:${tag}_c5"; do
  name=${pair%%:*}; dir=gpurun_out/${pair#*:}
  [ -d "$dir" ] || continue
  f=$(find "$dir/stats" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" profiles/${tag}_${name}_kernel_stats.csv
  cp "$dir/pmc_summary.txt" profiles/${tag}_${name}_pmc_summary.txt
  cp "$dir/bench_under_rocprof.json" profiles/${tag}_${name}_bench_under_rocprof.json
  python3 tools/mfma_clock_table.py "$dir" > profiles/${tag}_${name}_mfma_clock_table.txt
done
for w in c1 c2 c3 c4share c5s c5; do
  [ -f gpurun_out/${tag}_bench_$w.json ] && cp gpurun_out/${tag}_bench_$w.json profiles/${tag}_bench_${w}_extra.json
done
ls profiles | grep "^${tag}_" | wc -l
