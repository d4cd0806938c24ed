#!/bin/bash
# GPU box (round 6): soak of the end-to-end parity test over many more seeded random shapes than the suite runs (48):
# fit -> placement -> train -> predict -> reconstruct against the oracle, every failure listed (no -x)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r06_soak}; mkdir -p $out
SPR_TEST_SHAPE_SEEDS=${SEEDS:-1500} timeout -k 10 1100 python3 -m pytest tests/test_gpu_parity.py -q -k random_shapes_end_to_end -p no:cacheprovider > $out/soak.log 2>&1
tail -15 $out/soak.log
# ... and the SHARDED soak: 150 seeded random shapes row-sharded over three gloo ranks on this GPU (p2p exchange, the library's
# first-exchange trial, deferred launches by default), against the oracle on the whole matrix
if [ "${2:-}" = sharded ]; then
  SPR_TEST_SHARD_SEEDS=${SHARD_SEEDS:-150} timeout -k 10 1000 python3 -m pytest tests/test_dist_gpu_gloo.py -q -k random_shapes -p no:cacheprovider > $out/soak_sharded.log 2>&1
  tail -5 $out/soak_sharded.log
fi
