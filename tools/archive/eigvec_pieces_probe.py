#!/usr/bin/env python3
"""Host time of every piece of the top-r eigenvector route of fit() (sparse_sensing._eigh_tridiagonal / _eigvecs_top) on this
host, one BLAS thread: what is LAPACK, what is the library's batched inverse iteration, what is NumPy glue (copies, checks).
usage: python tools/eigvec_pieces_probe.py [m r ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.linalg import lapack
import openmeasure_amd.sparse_sensing as ss
from openmeasure_amd import _lib

lib = _lib.load()
args = [int(a) for a in sys.argv[1:]] or [256, 64, 512, 128]


def t(f, n=200):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        out = f()
    return 1e3 * (time.perf_counter() - t0) / n, out


for m, r in zip(args[::2], args[1::2]):
    rng = np.random.default_rng(m)
    A = rng.standard_normal((4 * m, m)) * (10 ** (-3 * np.arange(m) / max(r - 1, 1)))
    G = A.T @ A
    with ss._one_blas_thread():
        lw = int(lapack.dsytrd_lwork(m, lower=1)[0])
        rows = []
        dt, (c, d, e, tau, info) = t(lambda: lapack.dsytrd(G.T, lower=1, lwork=lw)); rows.append(('dsytrd (scipy wrapper)', dt))
        dt, (lam, info) = t(lambda: lapack.dsterf(d, e)); rows.append(('dsterf', dt))
        w = np.ascontiguousarray(lam[m - r:])
        Z = np.empty((m, r))
        dt, _ = t(lambda: lib.spr_host_tridiag_vectors(d.ctypes.data, e.ctypes.data, m, w.ctypes.data, r, Z.ctypes.data, 4)); rows.append(('spr_host_tridiag_vectors', dt))

        def check():
            E = Z.T @ Z
            E[np.diag_indices(r)] -= 1.0
            return np.abs(E).max(), E
        dt, (_, E) = t(check); rows.append(('Z^T Z - I, max', dt))
        dt, Zf = t(lambda: np.asfortranarray(Z - 0.5 * (Z @ E))); rows.append(('Z - Z E / 2, Fortran copy', dt))
        dt, cq = t(lambda: np.asfortranarray(c[1:, :m - 1])); rows.append(('copy of the reflectors (m-1 x m-1)', dt))
        dt, z1 = t(lambda: np.asfortranarray(Zf[1:])); rows.append(('copy of Z[1:]', dt))
        lwq = int(lapack.dormqr('L', 'N', cq, tau, z1, lwork=-1)[1][0])
        dt, (out, _, info) = t(lambda: lapack.dormqr('L', 'N', cq, tau, z1, lwork=lwq)); rows.append(('dormqr (scipy wrapper)', dt))
        dt, V = t(lambda: np.ascontiguousarray(np.vstack([Zf[:1], out])[:, ::-1])); rows.append(('vstack + flip + contiguous', dt))
        dt, _ = t(lambda: np.abs(V.T @ V - np.eye(r)).max()); rows.append(('V^T V - I, max', dt))
        dt_a, (lam2, fac) = t(lambda: ss._eigh_tridiagonal(G))
        dt_b, V2 = t(lambda: ss._eigvecs_top(fac, lam2, r))
    print(f'm = {m}, r = {r}:')
    for k, v in rows:
        print(f'    {k:40s} {v:7.3f} ms')
    print(f'    {"sum of the pieces":40s} {sum(v for _, v in rows):7.3f} ms')
    print(f'    product: _eigh_tridiagonal {dt_a:.3f} + _eigvecs_top {dt_b:.3f} = {dt_a + dt_b:.3f} ms')
