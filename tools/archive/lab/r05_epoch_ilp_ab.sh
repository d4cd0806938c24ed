# GPU box (round 5): full epoch sweep -- tiles one after the other (default) against the tiles' MFMA chains side by side (SPR_QR_EPOCH_ILP=1),
# both in the register-direct form (SPR_QR_EPOCH_STREAM=0), alternating in one call
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export SPR_QR_EPOCH_STREAM=0
SPR_QR_EPOCH_ILP=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "placement or pivots or pool or epoch or qr" 2>&1 | tail -3 || exit 1
for rep in 1 2; do
for nq in 16 32 48 64; do
  for s in 0 1; do SPR_QR_EPOCH_ILP=$s python3 tools/archive/lab/sweep_time.py 90000000 64 $nq 2>/dev/null | sed "s/^shipped/ilp=$s/"; done
done; done
for s in 0 1; do SPR_QR_EPOCH_ILP=$s python3 tools/archive/lab/sweep_time.py 100000000 64 48 f32 2>/dev/null | sed "s/^shipped/ilp=$s/"; done
