"""Host eigen-solve timing vs BLAS thread cap (run on the GPU box's host): the direct LAPACK `dsyevd` call of
`sparse_sensing._eigh_small` at the snapshot counts of the BASELINE configs and of the wide tests."""
import sys
import time

import numpy as np
from scipy.linalg import lapack
from threadpoolctl import ThreadpoolController

ctl = ThreadpoolController()
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024]
for m in sizes:
    A = np.random.default_rng(0).standard_normal((4 * m, m))
    G = A.T @ A
    for nt in (1, 2, 4, 8, 16, 32):
        with ctl.limit(limits=nt, user_api='blas'):
            lapack.dsyevd(G.T, lower=1)
            reps = 10 if m <= 512 else 3
            t = time.perf_counter()
            for _ in range(reps):
                w, v, info = lapack.dsyevd(G.T, lower=1)
            dt = (time.perf_counter() - t) / reps
        print(f'm={m} threads={nt}: {dt * 1e3:.3f} ms', flush=True)
