#!/bin/bash
# A/B of SPR.placement_norms inside the timed step (fit + reconstruct): does the norm epilogue of the projection cost the step
# anything?  Alternates the two settings on one box.   usage: tools/r04_norms_ab.sh [out-dir-name]
set -e -o pipefail
OUT=gpurun_out/${1:-r04_norms_ab}
mkdir -p $OUT
for i in 1 2 3; do
  for pn in auto off; do
    python3 bench.py --workload c4 --share-of 8 --share-rank 3 --steps 40 --warmup 8 --placement-norms $pn > $OUT/c4share_${pn}_$i.json 2> $OUT/err.log
    python3 bench.py --workload c3 --steps 10 --warmup 3 --placement-norms $pn > $OUT/c3_${pn}_$i.json 2>> $OUT/err.log
  done
done
python3 - "$OUT" <<'PY'
import json, sys, glob
for w in ('c4share', 'c3'):
    for pn in ('auto', 'off'):
        rows = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f'{sys.argv[1]}/{w}_{pn}_*.json'))]
        print(w, pn, 'ms_per_step', [r['ms_per_step'] for r in rows], 'project', [r['phases']['project']['ms'] for r in rows],
              'placement_ms', [r.get('placement_ms') for r in rows])
PY
