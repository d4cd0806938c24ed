"""How long does a small pageable H2D copy take next to a nearly full HBM? (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmeasure_amd.engine import HipEngine
eng = HipEngine()
def sync(): torch.cuda.synchronize()
def t_ms(fn, reps=5):
    out = []
    for _ in range(reps):
        sync(); t0 = time.perf_counter(); fn(); sync(); out.append(1e3 * (time.perf_counter() - t0))
    return ' '.join(f'{x:.2f}' for x in out)
W = np.random.default_rng(0).standard_normal((256, 64))
pinned = torch.empty((256, 64), dtype=torch.float64, pin_memory=True)
dst = torch.empty((256, 64), dtype=torch.float64, device='cuda')
def via_pinned():
    pinned.numpy()[...] = W
    dst.copy_(pinned, non_blocking=True)
for gb in (0, 100, 230):
    big = torch.empty(int(gb * 1e9) // 8 + 8, dtype=torch.float64, device='cuda')
    big[::4096].zero_(); sync()
    print(f'--- {gb} GB resident')
    print('to_device (fresh ndarray)   :', t_ms(lambda: eng.to_device(W / 1.0)))
    print('to_device (same ndarray)    :', t_ms(lambda: eng.to_device(W)))
    print('pinned staging + async copy :', t_ms(via_pinned))
    time.sleep(0.005)
    print('after 5 ms idle, fresh      :', t_ms(lambda: (time.sleep(0.005), eng.to_device(W / 1.0))))
    print('to_host 4.7 MB              :', t_ms(lambda: eng.to_host(big[:589824])))
    del big
    torch.cuda.empty_cache()
