"""GPU host: the one-call top-r route (spr_host_eig_top) against dsyevd and the Python-glued route (round 5)."""
import numpy as np, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import openmeasure_amd.sparse_sensing as ss
rng = np.random.default_rng(0)
for m, r in ((64, 32), (41, 14), (128, 32), (256, 64), (512, 128), (40, 20)):
    A = rng.standard_normal((8 * m, m)) * (0.97 ** np.arange(m)); A -= A.mean(axis=1, keepdims=True)
    G = A.T @ A
    def T(fn, reps=300):
        fn(); t0 = time.perf_counter()
        for _ in range(reps): o = fn()
        return 1e6 * (time.perf_counter() - t0) / reps, o
    tn, got = T(lambda: ss._eig_top_native(G, r))
    td, (w, vv) = T(lambda: ss._eigh_small(G))
    def top():
        la, fac = ss._eigh_tridiagonal(G); return la, ss._eigvecs_top(fac, la, r)
    tp, _ = T(top)
    if got is None:
        print(m, r, 'native: None', round(tn, 1), 'dsyevd', round(td, 1), 'python top-r', round(tp, 1)); continue
    lam, V = got
    print(m, r, 'native us', round(tn, 1), 'dsyevd', round(td, 1), 'python top-r', round(tp, 1), '| orth', np.abs(V.T @ V - np.eye(r)).max(), 'resid', np.abs(G @ V - V * lam[:r]).max() / lam[0])
