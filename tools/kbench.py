#!/usr/bin/env python3
"""Kernel micro-benchmark (GPU box): times each libspr_hip kernel with HIP events for a list of
shapes and prints ms, algorithmic GB/s and TFLOP/s.   python tools/kbench.py [cells,F,m,r ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.synth import make_R


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(1_000_000, 4, 64, 32), (1_000_000, 9, 256, 64)]
    eng = HipEngine()
    lib = eng.lib
    for cells, F, m, r in shapes:
        n = cells * F
        R = eng.to_device(make_R(m, r))
        X = eng.synth(n, m, 0, cells, R, 1e-3, 1)
        rowmean = eng.empty((n,)); fstats = eng.empty((F, 3)); gram = eng.empty((F, m, m))
        ws = eng._workspace('gram', lib.spr_stats_gram_workspace(m, F))
        st = eng._stream()
        t_g = timeit(lambda: lib.spr_stats_gram_f64(X.data_ptr(), n, m, m, 0, cells, F, 1, rowmean.data_ptr(), ws.data_ptr(), ws.numel(), st))
        t_f = timeit(lambda: lib.spr_stats_gram_finalize_f64(n, m, 0, cells, F, ws.data_ptr(), ws.numel(), fstats.data_ptr(), gram.data_ptr(), m, 0, st))
        W = eng.to_device(np.random.default_rng(0).standard_normal((m, r)))
        inv = eng.to_device(np.ones(F))
        Ur = eng.project(X, 0, cells, F, inv, W, rowmean=rowmean)
        t_p = timeit(lambda: eng.project(X, 0, cells, F, inv, W, out=Ur, rowmean=rowmean))
        a = eng.to_device(np.ones((1, r))); out = eng.empty((1, n))
        t_r = timeit(lambda: eng.reconstruct(Ur, 0, cells, F, rowmean, inv, a, out=out))
        qs = eng.qr_begin(Ur, 0, 8)
        eng.qr_step(qs, 0, qs['rec'][None], qs['tau'][None], True)
        t_q = timeit(lambda: eng.qr_refresh(qs, 0, 1))
        for j in range(1, 8):
            eng.qr_step(qs, j, qs['rec'][None], qs['tau'][None], True)
        t_q8 = timeit(lambda: eng.qr_refresh(qs, 0, 8))
        t_n = timeit(lambda: eng.qr_begin(Ur, 0, 8))
        xb = n * m * 8
        print(f'cells={cells} F={F} m={m} r={r}  X={xb / 1e9:.2f} GB')
        print(f'  stats_gram  {t_g:8.3f} ms  {xb / t_g / 1e6:8.1f} GB/s  {n * m * m / t_g / 1e9:7.2f} TF   (finalize {t_f:.3f} ms)')
        print(f'  project     {t_p:8.3f} ms  {(xb + n * r * 8) / t_p / 1e6:8.1f} GB/s  {2.0 * n * m * r / t_p / 1e9:7.2f} TF')
        print(f'  reconstruct {t_r:8.3f} ms  {(n * r * 8 + 16 * n) / t_r / 1e6:8.1f} GB/s')
        print(f'  qr_sweep x1 {t_q:8.3f} ms  {(n * r * 8 + 16 * n) / t_q / 1e6:8.1f} GB/s   x8 directions {t_q8:8.3f} ms  {(n * r * 8 + 16 * n) / t_q8 / 1e6:8.1f} GB/s   init(norms+cands) {t_n:8.3f} ms')
        del X, Ur, out, rowmean
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
