#!/bin/bash
# GPU box (round 6), N = 8 first-contact readiness on ONE GPU:
#  (a) config 5's per-rank HBM footprint at N = 8: one rank's share (204.8 GB f32 shard + 51.2 GB f32 basis), the p2p exchange with 7
#      imaginary peers (8 x 0.8 GB field copies = the 6.4 GB persistent copy of the gathered field a real rank holds) and a ballast
#      for what the all-gather leg adds on a real node (6.4 GB staging + 0.8 GB block + RCCL's own buffers): fit + placement + train
#      + predict + reconstruct complete, the line carries hbm.* (hipMemGetInfo) and peak_hbm_GB;
#  (b) the N-rank code path with SIX processes on this GPU (the box allows at most 6 processes on the card: 8 cannot be rehearsed
#      here) over gloo with the real IPC p2p exchange: 5 real peers dealt to 3 copy streams, the first-exchange trial, both exchanges
#      timed, per-rank timelines;
#  (c) the reproducibility stresses with deferred reconstructs (the default since this round).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r06_rehearsals}; mkdir -p $out
what=${2:-all}
if [ "$what" = all ] || [ "$what" = c5 ]; then
  timeout -k 10 900 python3 bench.py --workload c5 --share-of 8 --share-rank 3 --steps 3 --warmup 1 --no-cpu --p2p-loopback 7 --ballast-gb 10 > $out/c5_share8_ballast10.json 2> $out/c5_share8_ballast10.err || { echo "c5 rehearsal FAILED"; tail -8 $out/c5_share8_ballast10.err; exit 1; }
  python3 -c "
import json;d=json.load(open('$out/c5_share8_ballast10.json'));print('c5 share, loopback 7, ballast', d.get('ballast_GB'), 'GB:', d['ms_per_step'], 'ms/step; hbm', d['hbm'], 'peak_hbm_GB', d['peak_hbm_GB'], 'placement', d['placement_ms'], 'train', d['train_ms'], 'predict', d['predict_ms'])"
fi
if [ "$what" = all ] || [ "$what" = n6 ]; then
  for n in 6; do
    SPR_BENCH_ONE_GPU=1 SPR_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus $n --workload c4s --steps 5 --warmup 2 > $out/rehearsal_n$n.json 2> $out/rehearsal_n$n.err || { tail -30 $out/rehearsal_n$n.err; exit 1; }
    python3 -c "
import json;d=json.load(open('$out/rehearsal_n$n.json'));print('rehearsal n=$n', d['ms_per_step'], d['comm']['gather_path'], {k:(v.get('ms_per_step'),v.get('ms_per_step_sync_gather'),v.get('gather_ms'),v.get('gather_exposed_ms'),v.get('skipped')) for k,v in d['comm']['paths'].items()}, d['comm'].get('p2p_host_ms_per_gather'), d['comm'].get('p2p_copy_streams'), d.get('slowest_rank'))"
  done
fi
if [ "$what" = all ] || [ "$what" = stress ]; then
  timeout -k 10 600 python3 tools/step_stress.py 200 > $out/step_stress.txt 2>&1 || { tail -5 $out/step_stress.txt; exit 1; }
  cat $out/step_stress.txt
  timeout -k 10 400 python3 tools/p2p_stress.py 400 7 > $out/p2p_stress.txt 2>&1 || { tail -5 $out/p2p_stress.txt; exit 1; }
  grep defer_reconstruct $out/p2p_stress.txt
fi
