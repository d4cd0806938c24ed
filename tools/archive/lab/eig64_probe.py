"""GPU host: what the 64 x 64 eigen-solve between the passes of a config-2 fit() costs, piece by piece."""
import time
import numpy as np
from scipy.linalg import lapack
from threadpoolctl import ThreadpoolController
rng = np.random.default_rng(0)
for m in (41, 64, 128):
    A = rng.standard_normal((4000, m)); G = A.T @ A
    ctl = ThreadpoolController()
    def t(fn, n=300):
        fn(); t0 = time.perf_counter()
        for _ in range(n): fn()
        return 1e6 * (time.perf_counter() - t0) / n
    def plain(): return lapack.dsyevd(G.T, compute_v=1, lower=1)
    def limited():
        with ctl.limit(limits=1, user_api='blas'):
            return lapack.dsyevd(G.T, compute_v=1, lower=1)
    def only_limit():
        with ctl.limit(limits=1, user_api='blas'):
            pass
    def evr(): return lapack.dsyevr(G.T, compute_v=1, lower=1, range='I', il=m - m // 2 + 1, iu=m)
    libs = [l for l in ctl.lib_controllers if l.user_api == 'blas']
    def direct():
        old = [l.get_num_threads() for l in libs]
        for l in libs: l.set_num_threads(1)
        r = lapack.dsyevd(G.T, compute_v=1, lower=1)
        for l, o in zip(libs, old): l.set_num_threads(o)
        return r
    for l in libs: l.set_num_threads(1)
    one = t(plain)
    for l in libs: l.set_num_threads(64)
    print(f'm={m}: dsyevd default pool {t(plain):.1f} us | inside ctl.limit(1) {t(limited):.1f} | empty limit context {t(only_limit):.1f} | '
          f'direct set_num_threads(1)+restore {t(direct):.1f} | pool already at 1 thread {one:.1f} | dsyevr top half (pool 64) {t(evr):.1f} | np.linalg.eigh {t(lambda: np.linalg.eigh(G)):.1f}')
