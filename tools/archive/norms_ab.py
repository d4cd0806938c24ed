"""A/B of ROM.placement_norms on fit() wall time, alternating, at a bench workload's shard (GPU box).
usage: python tools/norms_ab.py [c3|c3s|c4s|c5s] [rows divisor]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

import bench  # noqa: E402
from openmeasure_amd.engine import HipEngine  # noqa: E402
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix  # noqa: E402
from openmeasure_amd.synth import make_R  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c3s'
div = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = bench.WORKLOADS[name]
eng = HipEngine('cuda:0')
F, m, s = wl['features'], wl['m'], wl['s']
n_points = wl['cells']
n = n_points * F // div
R = eng.to_device(make_R(m, s, seed=1234))
Xd = eng.synth(n, m, 0, n_points, R, 1e-3, 1234)
from openmeasure_amd.sparse_sensing import RowShard  # noqa: E402
spr = SPR(DeviceMatrix(Xd), F, None, engine=eng) if div == 1 else None
if spr is None:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
    dist.init_process_group('gloo', rank=0, world_size=1)
    spr = SPR(DeviceMatrix(Xd), F, None, shard=RowShard(0, n_points * F, partial=True), engine=eng)


def fit_ms(reps=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        spr.fit(select_modes='number', n_modes=s)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps


for mode in (None, False, None, False, True, False, None):
    spr.placement_norms = mode
    fit_ms(3)
    print(f'placement_norms={mode}: fit {fit_ms():.3f} ms  (norms written: {"nrm0" in spr._d})', flush=True)
