#!/bin/bash
# GPU box (round 6): the sharded GPU tests with ONE and TWO hardware queues per process (streams that share a hardware queue are
# served in submission order: a wait kernel submitted in front of the kernel that raises its counter would sit there until its
# time-out) -- the ready counter, the per-stream batches and the poison path under the tightest stream-to-queue mapping
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r06_hwq}; mkdir -p $out
for q in 1 2; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 500 python3 -m pytest tests/test_dist_gpu_gloo.py -x -q > $out/tests_q$q.log 2>&1; rc=$?
  echo "GPU_MAX_HW_QUEUES=$q: $(tail -1 $out/tests_q$q.log)"
  [ $rc -ne 0 ] && { tail -30 $out/tests_q$q.log; exit $rc; }
done
exit 0
