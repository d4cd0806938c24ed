#!/usr/bin/env python3
"""What the m x m host eigen-solve of fit() is made of on THIS host (one BLAS thread): dsyevd (what fit() calls) against the
pieces of a top-r route -- dsytrd, dsterf (all eigenvalues), dstemr (top r vectors of the tridiagonal matrix), dormqr (back
transformation of those r vectors) -- with the accuracy of the vectors that route returns."""
import time
import numpy as np
from scipy.linalg import lapack
from threadpoolctl import ThreadpoolController

ctl = ThreadpoolController()


def t(fn, reps=30):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return 1e3 * (time.perf_counter() - t0) / reps, out


for m, r in ((256, 64), (512, 128)):
    rng = np.random.default_rng(0)
    k = min(m, 2 * r)
    L = rng.standard_normal((8 * m, k)) * (10 ** (-3 / (r - 1))) ** np.arange(k)
    A = L @ rng.standard_normal((k, m)) + 1e-3 * rng.standard_normal((8 * m, m))
    A -= A.mean(axis=1, keepdims=True)
    G = A.T @ A
    with ctl.limit(limits=1, user_api='blas'):
        t_all, (w, v, info) = t(lambda: lapack.dsyevd(G.T, compute_v=1, lower=1))
        lw = int(lapack.dsytrd_lwork(m, lower=1)[0])
        t_trd, (c, d, e, tau, info) = t(lambda: lapack.dsytrd(G.T, lower=1, lwork=lw))
        t_erf, (wall, info) = t(lambda: lapack.dsterf(d, e))
        t_mr, res = t(lambda: lapack.dstemr(d, np.append(e, 0.0), 2, 0., 0., m - r + 1, m, compute_v=1))
        iblock = np.ones(m, dtype=np.int32); isplit = np.zeros(m, dtype=np.int32); isplit[0] = m
        wz = np.ascontiguousarray(wall[m - r:])
        t_in, (zz, info_in) = t(lambda: lapack.dstein(d, e, wz, iblock, isplit))
        Z = np.asfortranarray(zz[:, :r])
        cq = np.asfortranarray(c[1:, :m - 1])
        lwk = int(lapack.dormqr('L', 'N', cq, tau, np.asfortranarray(Z[1:]), lwork=-1)[1][0])
        t_bk, (out, _, _) = t(lambda: lapack.dormqr('L', 'N', cq, tau, np.asfortranarray(Z[1:]), lwork=lwk))
        V = np.vstack([Z[:1], out])
        ref = v[:, m - r:]
        print(f'm={m} r={r}: dsyevd {t_all:.3f} ms | dsytrd {t_trd:.3f} + dsterf {t_erf:.3f} + dstein {t_in:.3f} (dstemr {t_mr:.3f}) + dormqr {t_bk:.3f} = '
              f'{t_trd + t_erf + t_in + t_bk:.3f} ms | top-r vectors vs dsyevd {np.abs(np.abs(V.T @ ref) - np.eye(r)).max():.1e}, '
              f'orthogonality {np.abs(V.T @ V - np.eye(r)).max():.1e}', flush=True)
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import openmeasure_amd.sparse_sensing as ss
    t_full = t(lambda: ss._eigh_small(G))[0]
    t_top = t(lambda: ss._eigvecs_top(ss._eigh_tridiagonal(G)[1], ss._eigh_tridiagonal(G)[0], r))[0]
    t_tri = t(lambda: ss._eigh_tridiagonal(G))[0]
    print(f'   product functions: _eigh_small {t_full:.3f} ms | _eigh_tridiagonal {t_tri:.3f} + _eigvecs_top {t_top - t_tri:.3f} ms', flush=True)
    for nt in (2, 4, 8):
        with ctl.limit(limits=nt, user_api='blas'):
            print(f'   dsyevd with {nt} BLAS threads: {t(lambda: lapack.dsyevd(G.T, compute_v=1, lower=1))[0]:.3f} ms', flush=True)
