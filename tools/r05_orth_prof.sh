#!/bin/bash
# GPU box (round 5): kernel stats of optimal_placement with SPR_QR_ORTH_TILE = 0 | 1 (qr_step_fused_kernel / qr_orth_kernel durations)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_orth_prof}; mkdir -p $out
export TMPDIR=/tmp
for wl in ${WLS:-c3 c2}; do for tile in ${TILES:-0 1}; do
  export SPR_QR_ORTH_TILE=$tile
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${wl}_tile$tile -o p -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu > $out/${wl}_tile$tile.json 2> $out/${wl}_tile$tile.err || { tail -5 $out/${wl}_tile$tile.err; exit 1; }
  f=$(find $out/${wl}_tile$tile -name "*kernel_stats.csv" | head -1)
  [ -z "$f" ] && { echo "no kernel_stats.csv"; ls -R $out/${wl}_tile$tile | head; exit 1; }
  echo "== $wl tile=$tile"; grep -i "qr_step_fused\|qr_orth\|qr_cand_best" $f | awk -F'",' '{print substr($1,1,60), $2}'
  cp $f $out/${wl}_tile${tile}_kernel_stats.csv
  t=$(find $out/${wl}_tile$tile -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/placement_trace.py $t | tee $out/${wl}_tile${tile}_placement_trace.txt
  rm -rf $out/${wl}_tile$tile
done; done
