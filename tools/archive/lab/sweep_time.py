#!/usr/bin/env python3
"""Times ONE full epoch sweep (qr_epoch_sweep_kernel, all rows) and one refresh with made-up orthonormal directions, for
A/B runs of kernel variants selected with SPR_HIP_LIBRARY (ablations give wrong norms; only the times matter).
usage: python tools/archive/lab/sweep_time.py [rows] [r] [directions] [f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from openmeasure_amd.engine import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 90_000_000
r = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 32
f32 = len(sys.argv) > 4 and sys.argv[4] == 'f32'
eng = HipEngine('cuda:0')
Ur = torch.empty((n, r), dtype=torch.float32 if f32 else torch.float64, device='cuda:0')
for i0 in range(0, n, 1 << 24):
    Ur[i0:i0 + (1 << 24)].normal_()
st = eng.qr_begin(Ur, 0, r)
eng.qr_epoch_begin(st)
Q, _ = torch.linalg.qr(torch.randn(r, r, dtype=torch.float64, device='cuda:0'))
st['Q'][:nq] = Q.T[:nq]
st['piv'][:nq] = torch.arange(nq, device='cuda:0') * 1000
ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
for k in range(7):
    ev[k].record()
    eng.qr_epoch_sweep(st, 0, nq, 0)
ev[7].record()
torch.cuda.synchronize()
ts = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(2, 7))
gb = n * r * (4 if f32 else 8) / 1e9
print(f'{os.environ.get("SPR_HIP_LIBRARY", "shipped")}: rows {n} r {r} directions {nq} {"f32" if f32 else "f64"}: full epoch sweep '
      f'{ts[2]:.3f} ms median ({ts[0]:.3f} .. {ts[-1]:.3f}) = {gb / ts[2]:.2f} TB/s of the basis')
