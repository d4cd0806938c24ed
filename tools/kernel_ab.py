#!/usr/bin/env python3
"""Interleaved A/B of ONE kernel between builds of libspr_hip.so, in one process (GPU box).

    python tools/kernel_ab.py gram|project "cells,F,m,r" libA.so libB.so [...]  [--rounds 12]

Every build is loaded with its own ctypes handle; the rounds alternate A, B, ... on the same buffers and report the
median and the minimum per build (box-to-box and run-to-run spread is larger than most kernel-level effects: only
comparisons inside one process mean anything, cdna_hip_programming.md rule 24)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmeasure_amd import _lib
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.synth import make_R


def load(path):
    lib = C.CDLL(os.path.abspath(path), mode=C.RTLD_LOCAL)
    for name in ('spr_stats_gram_workspace', 'spr_stats_gram_f64', 'spr_project_norms_f64', 'spr_project_f64'):
        res, args = _lib.PROTOTYPES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    rounds = 12
    argv = sys.argv[1:]
    if '--rounds' in argv:
        i = argv.index('--rounds'); rounds = int(argv[i + 1]); del argv[i:i + 2]
    what, shape, paths = argv[0], argv[1], argv[2:]
    cells, F, m, r = (int(v) for v in shape.split(','))
    n = cells * F
    eng = HipEngine()
    libs = [load(p) for p in paths]
    X = eng.synth(n, m, 0, cells, eng.to_device(make_R(m, r)), 1e-3, 1)
    rowmean = eng.empty((n,))
    ws = eng._workspace('gram', libs[0].spr_stats_gram_workspace(m, F))
    st = eng._stream()
    W = eng.to_device(np.random.default_rng(0).standard_normal((m, r)))
    inv = eng.to_device(np.ones(F))
    Ur = eng.empty((n, r + (r & 1)))
    nrm = eng.empty((n,))

    def run(lib):
        if what == 'gram':
            rc = lib.spr_stats_gram_f64(X.data_ptr(), n, m, m, 0, cells, F, 1, rowmean.data_ptr(), ws.data_ptr(), ws.numel(), st)
        else:
            rc = lib.spr_project_norms_f64(X.data_ptr(), n, m, m, 0, cells, F, 1, inv.data_ptr(), rowmean.data_ptr(),
                                           W.data_ptr(), r, Ur.data_ptr(), Ur.stride(0), nrm.data_ptr(), st)
        assert rc == 0, rc

    if what != 'gram':
        run_g = libs[0].spr_stats_gram_f64(X.data_ptr(), n, m, m, 0, cells, F, 1, rowmean.data_ptr(), ws.data_ptr(), ws.numel(), st)
    ts = [[] for _ in libs]
    for lib in libs:
        run(lib)
    torch.cuda.synchronize()
    for rd in range(rounds):
        for k, lib in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(lib); run(lib); run(lib); e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 3)
    flops = float(n) * m * m if what == 'gram' else 2.0 * n * m * r
    print(f'{what} {cells}x{F}x{m} r={r}: {rounds} interleaved rounds of 3 launches')
    for p, t in zip(paths, ts):
        t = np.array(t[2:])
        print(f'  {os.path.basename(p):40s} median {np.median(t):8.3f} ms  min {t.min():8.3f} ms  ({flops / np.median(t) / 1e9:6.2f} TFLOP/s)')


if __name__ == '__main__':
    main()
