"""Where the top-r eigen route of fit() starts to pay on this host: dsyevd against dsytrd + dsterf + batched inverse iterations + dormqr
at small m (GPU host, round 4: m = 64 133 vs 142 us, m = 96 229 vs 221, m = 128 413 vs 347 -- hence _EIGH_TOP_MIN_M = 96)."""
import time, numpy as np, sys
sys.path.insert(0, ".")
import openmeasure_amd.sparse_sensing as ss
def T(fn, reps=300):
    fn(); t0=time.perf_counter()
    for _ in range(reps): out=fn()
    return 1e6*(time.perf_counter()-t0)/reps
for m, r in ((64,32),(41,14),(96,24),(128,32)):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((8*m, m)) * (0.8 ** np.arange(m)); A -= A.mean(axis=1, keepdims=True)
    G = A.T @ A
    def top():
        lam, fac = ss._eigh_tridiagonal(G)
        return ss._eigvecs_top(fac, lam, r)
    print(m, r, 'dsyevd us', round(T(lambda: ss._eigh_small(G)),1), 'top-r us', round(T(top),1))
