"""GPU host: spr_host_svd_top (dgebrd + dbdsdc + batched Golub-Kahan vectors + dormbr) against np.linalg.svd at the refinement's shapes,
by BLAS pool size (the calls run under threadpoolctl limits)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from threadpoolctl import threadpool_limits
from openmeasure_amd import sparse_sensing as S
rng = np.random.default_rng(0)
for m, r in ((256, 64), (512, 128), (128, 32)):
    V, _ = np.linalg.qr(rng.standard_normal((m, m)))
    d = np.concatenate([np.logspace(0, -6.1, r), 10 ** -6.6 * (1 + 0.1 * rng.random(m - r))])
    E = 1e-3 * rng.standard_normal((m, m))
    R = np.linalg.cholesky(np.eye(m) + 0.5 * (E + E.T)).T
    B = R * d[None, :]
    ptrs = S._lapack_svd_pointers()
    from openmeasure_amd import _lib
    lib = _lib.load()
    Sv, Vr = np.empty(m), np.empty((m, r))
    for th in (1, 2, 4, 8):
        with threadpool_limits(th):
            def nat():
                lib.spr_host_svd_top(B.ctypes.data, m, r, Sv.ctypes.data, Vr.ctypes.data, *ptrs)
            def full():
                np.linalg.svd(B)
            out = []
            for f in (nat, full):
                f()
                t0 = time.perf_counter()
                for _ in range(10):
                    f()
                out.append((time.perf_counter() - t0) / 10 * 1e3)
        print(f'm={m} r={r} threads={th}: svd_top {out[0]:.3f} ms   np.linalg.svd {out[1]:.3f} ms', flush=True)
