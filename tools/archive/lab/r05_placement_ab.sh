# GPU box: optimal_placement at the bench workloads, epoch-sweep variants alternating in one call (SPR_QR_EPOCH_ILP = 0 | 1 | 2)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
for wlargs in "--workload c3" "--workload c4 --share-of 8 --share-rank 3" "--workload c2"; do
for rep in 1 2; do for ilp in 0 1 2; do
  SPR_QR_EPOCH_ILP=$ilp python3 bench.py $wlargs --steps 3 --warmup 1 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ilp=$ilp', d['config']['workload'][:14], 'placement_ms', d['placement_ms'], 'sweeps', d['pivot_sweeps'], d['path']['pivot_pool_sweeps'], 'min gap', '%.3e' % d['min_pivot_gap'])"
done; done; done
