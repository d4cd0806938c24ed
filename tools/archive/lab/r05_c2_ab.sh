# GPU box: A/B of the small-transfer fast paths (download kernel + polled ticket, reusable W upload) at config 1 / 2, alternating in one call
for rep in 1 2 3; do for dl in 0 1; do for w in c2 c1; do SPR_DL_KERNEL=$dl python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dl=$dl $w', d['ms_per_step'], d['gaps_ms']['gram_to_project'], [v['ms'] for k,v in d['phases'].items() if k!='peaks'])"; done; done; done
