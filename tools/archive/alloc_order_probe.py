"""Does the position of the basis in HBM change the speed of a streaming sweep over it? (GPU box)
Times the MFMA refresh sweep (46 GB read) on a random 90M x 64 basis allocated before / after a 184 GB block."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmeasure_amd.engine import HipEngine
eng = HipEngine()
n, r = 90_000_000, 64
def make_U():
    U = torch.empty(n, r, dtype=torch.float64, device='cuda')
    for i0 in range(0, n, 10_000_000):
        U[i0:i0 + 10_000_000].normal_()
    return U.mul_(n ** -0.5)
def time_sweeps(U, label):
    st = eng.qr_begin(U, 0, r)
    eng.qr_step(st, 0, st['rec'][None], st['tau'][None], True)
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.qr_refresh(st, 0, 1); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    a = eng.to_device(np.ones((1, r))); rm = torch.zeros(n, dtype=torch.float64, device='cuda'); sc = eng.to_device(np.ones(1))
    tr = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.reconstruct(U, 0, n, 1, rm, sc, a); e1.record(); torch.cuda.synchronize(); tr.append(e0.elapsed_time(e1))
    print(f'{label:34s} refresh {np.median(ts):.2f} ms  reconstruct {np.median(tr):.2f} ms   U at {U.data_ptr():#x}')
    del st, rm
U = make_U()
time_sweeps(U, 'U alone')
big = torch.empty(184_320_000_000 // 8, dtype=torch.float64, device='cuda')
big[::512].fill_(1.0)
time_sweeps(U, 'U first, then 184 GB block')
del U
torch.cuda.empty_cache()
U2 = make_U()
time_sweeps(U2, 'U allocated after 184 GB block')
