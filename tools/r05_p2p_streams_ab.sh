#!/bin/bash
# GPU box (round 5): the loopback p2p exchange (7 imaginary peers) on one rank's block of config 4 at N = 8, by number of copy
# streams and number of hardware queues of the HIP process -- which combination lets the pushes run UNDER the next Gram pass.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/${1:-r05_p2p_ab}; mkdir -p $out
echo "hwq streams | pipelined ms | sync ms | gram project reconstruct | gather_ms exposed_ms" > $out/table.txt
python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 20 --warmup 5 --no-cpu --gather rccl 2> /dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rccl(1 rank) - |', d['ms_per_step'], '|', d['ms_per_step_sync_gather'], '|', *[v['ms'] for k,v in d['phases'].items() if k!='peaks'], '|', d['comm']['gather_ms'], d['comm']['gather_exposed_ms'])" >> $out/table.txt
for hwq in default 8 16; do
  for st in 7 3 2 1; do
    if [ $hwq = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hwq; fi
    SPR_P2P_STREAMS=$st python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 20 --warmup 5 --no-cpu --p2p-loopback 7 2> $out/err_${hwq}_$st.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$hwq $st |', d['ms_per_step'], '|', d['ms_per_step_sync_gather'], '|', *[v['ms'] for k,v in d['phases'].items() if k!='peaks'], '|', d['comm']['gather_ms'], d['comm']['gather_exposed_ms'])" >> $out/table.txt
  done
done
cat $out/table.txt
