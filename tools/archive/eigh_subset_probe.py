"""Host eigen-solve of the scaled Gram matrix: all pairs (dsyevd, what fit() calls) against the top-r pairs only through
SciPy's own drivers (dsyevr / dsyevx with an index range), one BLAS thread, on a Gram matrix with the bench spectrum."""
import sys
import time

import numpy as np
from scipy.linalg import eigh, lapack
from threadpoolctl import ThreadpoolController

ctl = ThreadpoolController()
for m, r in ((64, 32), (256, 64), (512, 128)):
    rng = np.random.default_rng(0)
    k = min(m, 2 * r)
    L = rng.standard_normal((8 * m, k)) * (10 ** (-3 / (r - 1))) ** np.arange(k)
    A = L @ rng.standard_normal((k, m)) + 1e-3 * rng.standard_normal((8 * m, m))
    A -= A.mean(axis=1, keepdims=True)
    G = A.T @ A
    with ctl.limit(limits=1, user_api='blas'):
        def t(fn, reps=20):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            return 1e3 * (time.perf_counter() - t0) / reps, out
        t_all, (w, v, info) = t(lambda: lapack.dsyevd(G.T, lower=1))
        res = {}
        for drv in ('evr', 'evx'):
            try:
                ms, (wr, vr) = t(lambda: eigh(G, subset_by_index=[m - r, m - 1], driver=drv, check_finite=False))
                err = np.abs(np.abs(vr.T @ v[:, m - r:]) - np.eye(r)).max()
                res[drv] = (ms, np.abs(wr - w[m - r:]).max() / w[-1], err)
            except Exception as e:      # noqa: BLE001
                res[drv] = repr(e)
        ms_vals, _ = t(lambda: lapack.dsterf(*lapack.dsytrd(G.T, lower=1)[1:3]) if False else eigh(G, eigvals_only=True, driver='ev', check_finite=False))
    print(f'm={m} r={r}: dsyevd all {t_all:.3f} ms | eigvals only {ms_vals:.3f} ms | ' +
          ' | '.join(f'{d}: {x[0]:.3f} ms (dlam {x[1]:.1e}, vec {x[2]:.1e})' if isinstance(x, tuple) else f'{d}: {x}' for d, x in res.items()), flush=True)
