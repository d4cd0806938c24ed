# GPU box (round 5): full epoch sweep, register-direct form (SPR_QR_EPOCH_STREAM=0) against the streaming form with the per-wave LDS
# transpose (default), alternating in one call; then the placement tests and the bench placement line.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "placement or pivots or pool or epoch or qr" 2>&1 | tail -3 || exit 1
for rep in 1 2; do
for nq in 16 32 48; do
  for s in 0 1; do SPR_QR_EPOCH_STREAM=$s python3 tools/archive/lab/sweep_time.py 90000000 64 $nq 2>/dev/null | sed "s/^shipped/stream=$s/"; done
done; done
for s in 0 1; do SPR_QR_EPOCH_STREAM=$s python3 tools/archive/lab/sweep_time.py 45000000 32 16 2>/dev/null | sed "s/^shipped/stream=$s/"; done
for s in 0 1; do SPR_QR_EPOCH_STREAM=$s python3 tools/archive/lab/sweep_time.py 100000000 64 32 f32 2>/dev/null | sed "s/^shipped/stream=$s/"; done
