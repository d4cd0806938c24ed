#!/bin/bash
# GPU box (round 6): device timeline of ONE pipelined step of rank 3's block of config 4 at N = 8 with the loopback p2p exchange,
# round-6 protocol (deferred launch in fit()'s host gap, ready counter, per-stream wait / arrive kernels): kernel + memory-copy trace
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r06_timeline; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -o run -- python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 8 --warmup 3 --no-cpu --p2p-loopback 7 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 tools/timeline.py $out/trace --step 6 > $out/timeline.txt 2>&1
head -70 $out/timeline.txt | cut -c1-160
