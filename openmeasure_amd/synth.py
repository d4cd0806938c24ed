"""Host side of the synthetic snapshot generator: the small k x (m+1)
right factor R with a designed spectrum.  The n x m matrix itself is produced on the
device by spr_synth_f64 (csrc/synth.hip)."""
import numpy as np


def make_R(m, s, seed=1234, ratio=1e3, extra_cols=1):
    """R = diag(rho^j) N(0,1), k = min(m, 2 s) rows, m + extra_cols columns (the extra column is
    the held-out state used for measurements); rho chosen so sigma_1/sigma_s ~ ratio."""
    k = min(m, 2 * s)
    rho = ratio ** (-1.0 / max(s - 1, 1))
    rng = np.random.default_rng(seed)
    return (rho ** np.arange(k))[:, None] * rng.standard_normal((k, m + extra_cols))
