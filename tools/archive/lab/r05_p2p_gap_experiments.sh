run() { label=$1; shift; env "$@" python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 20 --warmup 5 --no-cpu --p2p-loopback 7 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'], d['ms_per_step_sync_gather'], d['gaps_ms'], d['comm']['gather_ms'], d['comm']['gather_exposed_ms'])"; }
run base X=1
run norelease SPR_P2P_EXPERIMENT=norelease   # a measurement-only switch of p2p.py at the commit of the record (pushes did not wait for the peers' release); removed since
run blit SPR_P2P_BLIT=1
run hwq16 GPU_MAX_HW_QUEUES=16
run nullstream SPR_BENCH_STREAM=0
run nullstream_hwq16 SPR_BENCH_STREAM=0 GPU_MAX_HW_QUEUES=16
