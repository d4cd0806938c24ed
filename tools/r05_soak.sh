#!/bin/bash
# GPU box (round 5): soak of the end-to-end parity test over many more seeded random shapes than the suite runs (48):
# fit -> placement -> train -> predict -> reconstruct against the oracle, every failure listed (no -x)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_soak}; mkdir -p $out
SPR_TEST_SHAPE_SEEDS=${SEEDS:-1500} timeout -k 10 1100 python3 -m pytest tests/test_gpu_parity.py -q -k random_shapes_end_to_end -p no:cacheprovider > $out/soak.log 2>&1
tail -15 $out/soak.log
