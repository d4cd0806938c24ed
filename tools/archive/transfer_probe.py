#!/usr/bin/env python3
"""Host <-> device rates for the two big transfers of the reference's contract (GPU box): a host ndarray X handed to SPR()
(H2D, sparse_sensing.py:74 keeps the caller's array) and the (n, n_p) ndarray reconstruct() returns (D2H, :371-375).
Methods: torch's pageable copy; a pinned result tensor filled by one async copy; chunked staging through two pinned buffers
with 1 / 4 copy threads; hipHostRegister of the caller's array in place."""
import os, sys, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

dev = torch.device('cuda:0')
GB = 1e9


def sync():
    torch.cuda.synchronize()


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        sync(); t0 = time.perf_counter(); out = fn(); sync(); ts.append(time.perf_counter() - t0)
    return min(ts), out


# ---------------- D2H: a 720 MB field (config 3, one vector) ----------------
n = 90_000_000
d = torch.randn(n, dtype=torch.float64, device=dev)
t, _ = best(lambda: d.cpu().numpy())
print(f'D2H pageable t.cpu()            : {n * 8 / t / GB:6.1f} GB/s ({1e3 * t:.1f} ms)', flush=True)


def d2h_pinned_fresh():
    h = torch.empty(n, dtype=torch.float64, pin_memory=True)
    h.copy_(d, non_blocking=True)
    sync()
    return h.numpy()


t0 = time.perf_counter(); a = d2h_pinned_fresh(); t_first = time.perf_counter() - t0
del a
t, a = best(d2h_pinned_fresh)
print(f'D2H into a pinned result tensor : {n * 8 / t / GB:6.1f} GB/s ({1e3 * t:.1f} ms; first call, pinning {1e3 * t_first:.0f} ms)', flush=True)
keep = a                                           # the caller still holds the previous result: a second block gets pinned
t, b = best(d2h_pinned_fresh)
print(f'  ... while the previous result is alive: {n * 8 / t / GB:6.1f} GB/s ({1e3 * t:.1f} ms)', flush=True)
del keep, a, b

stage = [torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
side = torch.cuda.Stream()


def d2h_staged(threads):
    out = np.empty(n, dtype=np.float64)
    ch = (64 << 20) // 8
    evs = [None, None]
    pool = ThreadPoolExecutor(threads) if threads > 1 else None
    pend = []
    k = 0
    for i0 in range(0, n, ch):
        i1 = min(n, i0 + ch)
        s = stage[k & 1][:(i1 - i0) * 8].view(torch.float64)
        if evs[k & 1] is not None:                 # the slot's previous host copy must be done before the DMA overwrites it
            for f in pend[k & 1]:
                f.result()
        with torch.cuda.stream(side):
            s.copy_(d[i0:i1], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(side)
        ev.synchronize()
        src = s.numpy()
        if pool is None:
            out[i0:i1] = src
            fs = []
        else:
            q = (i1 - i0 + threads - 1) // threads
            fs = [pool.submit(np.copyto, out[i0 + j * q:min(i1, i0 + (j + 1) * q)], src[j * q:min(i1 - i0, (j + 1) * q)]) for j in range(threads)]
        while len(pend) < 2:
            pend.append([])
        pend[k & 1] = fs
        evs[k & 1] = ev
        k += 1
    for fs in pend:
        for f in fs:
            f.result()
    if pool:
        pool.shutdown()
    return out


for th in (1, 4):
    t, _ = best(lambda: d2h_staged(th), reps=2)
    print(f'D2H staged 2 x 64 MiB, {th} thread(s): {n * 8 / t / GB:6.1f} GB/s ({1e3 * t:.1f} ms)', flush=True)
del d

# ---------------- H2D: a 2 GB host ndarray (config 2) ----------------
X = np.random.default_rng(0).standard_normal((4_000_000, 64))
nb = X.nbytes
t, _ = best(lambda: torch.as_tensor(X).to(dev))
print(f'H2D pageable .to(device)        : {nb / t / GB:6.1f} GB/s ({1e3 * t:.1f} ms)', flush=True)
Xd = torch.empty(X.shape, dtype=torch.float64, device=dev)
flat_d = Xd.view(-1)
flat_h = X.reshape(-1)


def h2d_staged(threads):
    ch = (64 << 20) // 8
    pool = ThreadPoolExecutor(threads) if threads > 1 else None
    evs = [None, None]
    k = 0
    for i0 in range(0, flat_h.size, ch):
        i1 = min(flat_h.size, i0 + ch)
        s = stage[k & 1][:(i1 - i0) * 8].view(torch.float64)
        if evs[k & 1] is not None:
            evs[k & 1].synchronize()               # the DMA that read this slot last has finished
        dst = s.numpy()
        if pool is None:
            dst[:] = flat_h[i0:i1]
        else:
            q = (i1 - i0 + threads - 1) // threads
            for f in [pool.submit(np.copyto, dst[j * q:min(i1 - i0, (j + 1) * q)], flat_h[i0 + j * q:min(i1, i0 + (j + 1) * q)]) for j in range(threads)]:
                f.result()
        with torch.cuda.stream(side):
            flat_d[i0:i1].copy_(s, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(side)
        evs[k & 1] = ev
        k += 1
    side.synchronize()
    if pool:
        pool.shutdown()


for th in (1, 4, 8):
    t, _ = best(lambda: h2d_staged(th), reps=2)
    print(f'H2D staged 2 x 64 MiB, {th} thread(s): {nb / t / GB:6.1f} GB/s ({1e3 * t:.1f} ms)', flush=True)
assert torch.equal(Xd.cpu(), torch.as_tensor(X))


def h2d_registered():
    rt = torch.cuda.cudart()
    rc = rt.cudaHostRegister(X.ctypes.data, nb, 0)
    assert int(rc) == 0, rc
    Xd.copy_(torch.as_tensor(X), non_blocking=True)
    sync()
    rt.cudaHostUnregister(X.ctypes.data)


try:
    t, _ = best(h2d_registered, reps=2)
    print(f'H2D hipHostRegister in place    : {nb / t / GB:6.1f} GB/s ({1e3 * t:.1f} ms incl. register + unregister)', flush=True)
except Exception as e:     # noqa: BLE001
    print('H2D hipHostRegister in place    : failed:', repr(e)[:200], flush=True)
