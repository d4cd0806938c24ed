#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_push_probe; mkdir -p $out
for cfg in "7 3 0" "7 3 60" "7 3 200"; do
  echo "== peers/streams/busy $cfg" >> $out/probe2.txt
  timeout -k 10 120 python3 tools/p2p_push_probe.py $cfg >> $out/probe2.txt 2>&1 || exit 1
done
SPR_P2P_PROBE=1 timeout -k 10 120 python3 tools/p2p_push_probe.py 7 3 200 2>&1 | grep "p2p probe" | tail -28 >> $out/probe2.txt
SPR_P2P_PROBE=1 timeout -k 10 200 python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 6 --warmup 2 --no-cpu --p2p-loopback 7 > $out/bench_probe.json 2> $out/bench_probe.err
grep "p2p probe" $out/bench_probe.err | tail -42 >> $out/probe2.txt
cat $out/probe2.txt
