timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for mode in 0 1; do
  SPR_GRAM_OWN=$mode timeout -k 10 300 python bench.py --workload c5s --no-cpu --steps 8 --warmup 2 2> /dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c5s own=$mode',d['ms_per_step'],{k:v['ms'] for k,v in d['phases'].items()})"
done
for mode in 0 1; do
  SPR_GRAM_OWN=$mode timeout -k 10 300 python bench.py --workload c5 --no-cpu --steps 4 --warmup 1 2> /dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('c5 own=$mode',d['ms_per_step'],{k:v['ms'] for k,v in d['phases'].items()})"
done
