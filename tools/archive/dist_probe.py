"""Where does the collective path of fit() lose time?  One-rank RCCL group at c3s, host and GPU timelines. (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix, RowShard
import openmeasure_amd.sparse_sensing as ss
from openmeasure_amd.synth import make_R
eng = HipEngine()
cells, F, m, s = int(os.environ.get("CELLS", 1_000_000)), 9, 256, 64
R = eng.to_device(make_R(m, s))
Xd = eng.synth(cells * F, m, 0, cells, R, 1e-3, 1234)
def sync(): torch.cuda.synchronize()
spr = SPR(DeviceMatrix(Xd), F, None, shard=RowShard(0, cells * F, force_collectives=True), engine=eng)
marks, cur = [], {}
def gpu_mark(label):
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); marks.append((label, e))
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        gpu_mark(label + '<'); t0 = time.perf_counter(); r_ = f(*a, **k)
        cur[label] = cur.get(label, 0.0) + 1e3 * (time.perf_counter() - t0); gpu_mark(label + '>')
        return r_
    setattr(obj, name, g)
wrap(eng, 'to_host', 'd2h'); wrap(ss, '_eigh_small', 'eigh'); wrap(eng, 'stats_gram', 'gram'); wrap(eng, 'project', 'proj')
wrap(spr, '_all_reduce', 'allred'); wrap(spr, '_all_gather', 'allgat'); wrap(dist, 'broadcast', 'bcast'); wrap(eng, 'to_device', 'h2d')
for _ in range(3):
    spr.fit(select_modes='number', n_modes=s)
sync()
import gc
gc_log = []
def gc_cb(phase, info):
    if phase == 'start': gc_log.append([info['generation'], time.perf_counter(), None])
    else: gc_log[-1][2] = 1e3 * (time.perf_counter() - gc_log[-1][1])
gc.callbacks.append(gc_cb)
for rep in range(16):
    if rep == 8:
        gc.collect(); gc.freeze(); print('gc.freeze() done')
    gc_log.clear()
    marks.clear(); cur.clear()
    sync(); t0 = time.perf_counter(); spr.fit(select_modes='number', n_modes=s); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    tl = ' '.join(f'{marks[i + 1][0]}+{marks[i][1].elapsed_time(marks[i + 1][1]):.1f}' for i in range(len(marks) - 1))
    print(f'fit {1e3*(t2-t0):6.1f} (host {1e3*(t1-t0):.1f}) host: ' + ' '.join(f'{a}={b:.1f}' for a, b in cur.items()))
    print('    gc: ' + str([(g, round(d, 1)) for g, _, d in gc_log if d and d > 0.5]))
dist.destroy_process_group()
