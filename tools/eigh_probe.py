"""Host eigen-solve timing vs BLAS thread cap (run on the GPU box's host)."""
import time
import numpy as np
from threadpoolctl import ThreadpoolController
ctl = ThreadpoolController()
for m in (64, 256, 512):
    A = np.random.default_rng(0).standard_normal((4 * m, m)); G = A.T @ A
    for nt in (1, 2, 4, 8, 16, 32):
        with ctl.limit(limits=nt, user_api='blas'):
            np.linalg.eigh(G)
            t = time.perf_counter()
            for _ in range(10):
                np.linalg.eigh(G)
            print(f'm={m} threads={nt}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms')
