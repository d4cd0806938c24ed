#!/usr/bin/env python3
"""GPU box (round 5): what a copy-engine transfer in flight costs the kernels it is meant to hide under.

One rank's block of BASELINE config 4 at N = 8 (11.25M rows x 256, f64).  The Gram pass and the projection are timed with HIP
events alone, then with 630 MB -- what a rank receives per step at N = 8 -- moving at the same time
  (a) device-to-device through the SDMA engines (spr_p2p_copy = hipMemcpyDeviceToDeviceNoCU): seven 90 MB pieces on seven
      streams, the shape of the p2p field exchange; on one GPU source and destination share the HBM, i.e. twice the HBM traffic
      of a real exchange, where one end is another GPU,
  (b) device-to-host into page-locked memory (the stand-in used for the host contract of reconstruct()),
  (c) the same 630 MB as the default device-to-device copy (blit kernels on compute units), for contrast.
Output: one table; profiles/r05_p2p_overlap_probe.txt is a copy of it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from openmeasure_amd import _lib  # noqa: E402
from openmeasure_amd.engine import HipEngine  # noqa: E402
from openmeasure_amd.synth import make_R  # noqa: E402

eng = HipEngine('cuda:0')
lib = eng.lib
n_points, F, m, r = 10_000_000, 9, 256, 64
n_loc, row0 = 11_250_000, 33_750_000
R = eng.to_device(make_R(m, r, seed=1234))
X = eng.synth(n_loc, m, row0, n_points, R, 1e-3, 1234)
W = eng.to_device(np.linalg.qr(np.random.default_rng(0).standard_normal((m, r)))[0])
inv = eng.to_device(np.ones(F))
PIECE = 90_000_000
src = torch.empty(7 * PIECE, dtype=torch.uint8, device=eng.device)
dst = torch.empty(7 * PIECE, dtype=torch.uint8, device=eng.device)
host = torch.empty(7 * PIECE, dtype=torch.uint8, pin_memory=True)
streams = [torch.cuda.Stream(eng.device) for _ in range(7)]
main = torch.cuda.current_stream(eng.device)


ROUNDS = dict(sdma=1, sdma16=16, blit=1, d2h=1, d2h_nocu=1)   # sdma16: a stress far beyond what a step moves (10 GB per Gram pass)


def copies(kind):
    """enqueue ROUNDS x 630 MB on the side streams; -> (start events, end events)"""
    e0s, e1s = [], []
    for j, s in enumerate(streams):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(s)
            a, b = src[j * PIECE:(j + 1) * PIECE], dst[j * PIECE:(j + 1) * PIECE]
            for _ in range(ROUNDS[kind]):
                if kind in ('sdma', 'sdma16'):
                    _lib.check(lib.spr_p2p_copy(b.data_ptr(), a.data_ptr(), PIECE, s.cuda_stream), 'spr_p2p_copy')
                elif kind == 'd2h_nocu':
                    _lib.check(lib.spr_p2p_copy(host[j * PIECE:(j + 1) * PIECE].data_ptr(), a.data_ptr(), PIECE, s.cuda_stream),
                               'spr_p2p_copy')
                elif kind == 'blit':
                    b.copy_(a, non_blocking=True)
                elif kind == 'd2h':
                    host[j * PIECE:(j + 1) * PIECE].copy_(a, non_blocking=True)
            e1.record(s)
        e0s.append(e0); e1s.append(e1)
    return e0s, e1s


def run(kernel, kind, reps=int(os.environ.get('PROBE_REPS', '8'))):
    ks, cs = [], []
    for _ in range(reps + 2):
        torch.cuda.synchronize()
        ev = torch.cuda.Event(); ev.record(main)
        for s in streams:
            s.wait_event(ev)
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if kind != 'none':
            c0, c1 = copies(kind)
        k0.record(main)
        if kernel == 'gram':
            rm, fs, g = eng.stats_gram(X, row0, n_points, F, center=True)
        else:
            u = eng.project(X, row0, n_points, F, inv, W, center=True, rowmean=rowmean, out=ur[0])
            ur[0] = u
        k1.record(main)
        torch.cuda.synchronize()
        ks.append(k0.elapsed_time(k1))
        if kind != 'none':
            cs.append(max(c0[0].elapsed_time(e) for e in c1))
    ks, cs = ks[2:], cs[2:]
    return float(np.median(ks)), float(np.min(ks)), (float(np.median(cs)) if cs else None)


rowmean, _, _ = eng.stats_gram(X, row0, n_points, F, center=True)
ur = [None]
print(f'one rank block of config 4 at N = 8: {n_loc} rows x {m} f64 ({n_loc * m * 8 / 1e9:.2f} GB); 630 MB = 7 x 90 MB in flight on 7 streams')
print(f'{"kernel":12s} {"copy in flight":34s} {"kernel ms (median / min)":26s} {"copies: first start -> last end, ms":36s} slow-down')
base = {}
for kernel in ('gram', 'project'):
    for kind, label in (('none', 'none'), ('sdma', 'D2D NoCU (SDMA), same GPU'), ('sdma16', 'D2D NoCU, 16 rounds (stress)'),
                        ('d2h', 'D2H pinned, torch copy_'), ('d2h_nocu', 'D2H pinned, NoCU (SDMA)'),
                        ('blit', 'D2D default (blit kernels)'), ('none', 'none (again)')):
        med, mn, c = run(kernel, kind)
        base.setdefault(kernel, med)
        gbs = '' if c is None else f'{c:8.3f}  ({ROUNDS[kind] * 7 * PIECE / c / 1e6:6.1f} GB/s, {ROUNDS[kind]} x 630 MB)'
        print(f'{kernel:12s} {label:34s} {med:9.3f} / {mn:9.3f}      {gbs:36s} {100 * (med / base[kernel] - 1):+5.1f} %')
