#!/usr/bin/env python3
"""Host algebra of one refinement pass of fit() (sparse_sensing._refine_spectrum) on THIS host, piece by piece:
eigh(H), M = L^1/2 Z^T D V^T, svd(M), eigvalsh of the retained block -- one BLAS thread against the default pool."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import openmeasure_amd.sparse_sensing as ss
from scipy.linalg import lapack


def t(fn, reps=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return 1e3 * (time.perf_counter() - t0) / reps, out


for m, r in ((256, 64), (512, 128)):
    rng = np.random.default_rng(0)
    V, _ = np.linalg.qr(rng.standard_normal((m, m)))
    d = np.logspace(0, -9, m)
    E = 1e-5 * rng.standard_normal((m, m))
    H = np.eye(m) + E + E.T
    for label, ctx in (('1 thread', ss._one_blas_thread), ('default pool', __import__('contextlib').nullcontext)):
        with ctx():
            t_eig, (lamH, Z) = t(lambda: ss._eigh_small(H))
            t_M, M = t(lambda: (np.sqrt(np.maximum(lamH, 0.0))[:, None] * Z.T) * d[None, :] @ V.T)
            t_svd, _ = t(lambda: np.linalg.svd(M))
            t_sv, _ = t(lambda: np.linalg.svd(M, compute_uv=False))
            t_jsv, _ = t(lambda: lapack.dgejsv(M, joba=4, jobu=3, jobv=0))
            dn = np.sqrt(np.diag(H))
            t_ev, _ = t(lambda: np.linalg.eigvalsh((H / dn[:, None] / dn[None, :])[:r, :r]))
        print(f'm={m} {label:13s}: eigh(H) {t_eig:.2f} | M {t_M:.2f} | svd(M) {t_svd:.2f} (values only {t_sv:.2f}, dgejsv V only {t_jsv:.2f}) | eigvalsh {t_ev:.2f} ms', flush=True)
