"""GPU host: does a small BLAS pool help the tridiagonalisation (the largest piece of the host eigen-solve of fit())?"""
import time
import numpy as np
from scipy.linalg import lapack
from threadpoolctl import ThreadpoolController
ctl = ThreadpoolController()
rng = np.random.default_rng(0)
for m in (256, 512):
    A = rng.standard_normal((4 * m, m)); G = A.T @ A
    lw = int(lapack.dsytrd_lwork(m, lower=1)[0])
    row = []
    for nt in (1, 2, 4, 8, 16):
        with ctl.limit(limits=nt, user_api='blas'):
            lapack.dsytrd(G.T, lower=1, lwork=lw)
            t0 = time.perf_counter()
            for _ in range(100): lapack.dsytrd(G.T, lower=1, lwork=lw)
            row.append(f'{nt} threads {1e3 * (time.perf_counter() - t0) / 100:.3f} ms')
    print(f'dsytrd m = {m}: ' + ' | '.join(row))
