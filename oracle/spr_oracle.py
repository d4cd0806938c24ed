"""CPU oracle for the SPR hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy/SciPy restatement of the arithmetic that OpenMEASURE's
``openmeasure.sparse_sensing.ROM`` / ``SPR`` classes perform on the path

    scale_data('std', axis_cnt=1) -> thin SVD -> mode truncation -> QR-pivot
    sensor selection -> Theta = C.Ur -> scale_vector -> OLS predict -> reconstruct

Every function cites the reference lines it follows (paths relative to
``/root/reference/src/openmeasure/sparse_sensing.py``).  It is written as free
functions on plain arrays (no class state) so that a test can call one stage at a
time.

Who may import this module: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- there only as the checker / the timed CPU
baseline.  Nothing under ``openmeasure_amd/`` imports it; the product path has no
CPU fallback.

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against the
fixtures in ``tests/golden/*.npz``, which were produced by the *unmodified*
reference module imported in the build container by ``oracle/make_golden.py``
(the script, seeds and shapes are committed next to the fixtures).  The same
LAPACK entry points are used as in the reference (``dgesdd`` through
``np.linalg.svd``, ``dgeqp3`` through ``scipy.linalg.qr(pivoting=True)``, the
SVD-based ``np.linalg.pinv``), so on the fixture inputs the oracle reproduces the
reference bit for bit, not merely to tolerance.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


# ----------------------------------------------------------------------------
# a1  ROM.__init__ validation                      sparse_sensing.py:69-81
# ----------------------------------------------------------------------------
def check_inputs(X, n_features):
    """Argument checks of ``ROM.__init__`` (:69-72, :78-81); returns n_points."""
    if type(X) is not np.ndarray:
        raise TypeError('The matrix X is not a numpy array.')
    if type(n_features) is not int:
        raise TypeError('The parameter n_features is not an integer.')
    n = X.shape[0]
    if n % n_features != 0:
        raise Exception('The number of rows of X is not a multiple of n_features')
    return n // n_features


# ----------------------------------------------------------------------------
# a2  ROM.scale_data('std', axis_cnt=1)            sparse_sensing.py:106-171
# ----------------------------------------------------------------------------
def feature_scale(x, scale_type):
    """The per-feature scaling factor of ``ROM.scale_data`` for one feature block x
    (:114-161; the kurtosis-based 'vast_2/3/4' branches assign an m-vector to a column
    slice in the reference and only run when n_points == m, they are not restated)."""
    if scale_type == 'std':
        return np.std(x)                                  # :115
    if scale_type == 'none':
        return 1.                                         # :118
    if scale_type == 'pareto':
        return np.sqrt(np.std(x))                         # :121
    if scale_type == 'vast':
        return np.std(x) ** 2 / np.average(x)             # :124
    if scale_type == 'range':
        return np.max(x) - np.min(x)                      # :128
    if scale_type == 'level':
        return np.average(x)                              # :132
    if scale_type == 'max':
        return np.max(x)                                  # :135
    if scale_type == 'variance':
        return np.var(x)                                  # :138
    if scale_type == 'median':
        return np.median(x)                               # :141
    if scale_type == 'poisson':
        return np.sqrt(np.average(x))                     # :144
    if scale_type == 'l2-norm':
        return np.linalg.norm(x)                          # :160
    raise NotImplementedError('The scaling method selected has not been implemented yet')   # :164


def scale_data(X, n_features, scale_type='std', axis_cnt=1):
    """Row centring + one scaling factor per feature block.

    :112  X_cnt[block] = np.average(block, axis=axis_cnt)   (row mean, or the block's
          scalar mean when axis_cnt is None)
    :114-161  X_scl[block] = one scalar per feature (population std for 'std')
    :169  X0 = (X - X_cnt) / X_scl
    Returns (X_cnt (n,1), X_scl (n,1), X0 (n,m)), all float64 like the reference's
    ``np.zeros`` outputs (:106-107).
    """
    n = X.shape[0]
    n_points = n // n_features
    X_cnt = np.zeros((n, 1))
    X_scl = np.zeros((n, 1))
    for f in range(n_features):
        blk = slice(f * n_points, (f + 1) * n_points)
        x = X[blk, :]
        X_cnt[blk, 0] = np.average(x, axis=axis_cnt)
        X_scl[blk, 0] = feature_scale(x, scale_type)
    X0 = (X - X_cnt) / X_scl
    return X_cnt, X_scl, X0


def scale_data_std(X, n_features, axis_cnt=1):
    """``scale_data('std', axis_cnt)`` -- the default branch (:115)."""
    return scale_data(X, n_features, 'std', axis_cnt)


# ----------------------------------------------------------------------------
# a4  ROM.reduction                                sparse_sensing.py:314-338
# ----------------------------------------------------------------------------
def scale_limits(limits, X_cnt, X_scl, n_features):
    """ROM.scale_limits (:173-210): per feature (limit_f - X_cnt)/X_scl on that feature's rows; a block whose
    minimum falls below -1000 becomes the scalar -1000, else one whose maximum exceeds 1000 the scalar 1000 (:200-203)."""
    n = X_cnt.shape[0]
    n_points = n // n_features
    out = []
    for limit in limits:
        limit0 = np.zeros((n,))
        for i in range(n_features):
            sl = slice(i * n_points, (i + 1) * n_points)
            temp = (limit[i] - X_cnt[sl, 0]) / X_scl[sl, 0]
            if np.min(temp) < -1000:
                temp = -1000
            elif np.max(temp) > 1000:
                temp = 1000
            limit0[sl] = temp
        out.append(limit0)
    return out


def select_rank(exp_variance, n_cols, select_modes, n_modes):
    """How many modes survive -- the integer logic of ``ROM.reduction``.

    'variance' (:314-324): range check, 100 keeps everything, otherwise the first r
    with exp_variance[r-1] >= n_modes.  'number' (:326-331): type then range check.
    Anything else: ValueError (:333).
    """
    if select_modes == 'variance':
        if not 0 <= n_modes <= 100:
            raise ValueError('The parameter n_modes is outside the[0-100] range.')
        if n_modes == 100:
            return n_cols
        r = 1
        while exp_variance[r - 1] < n_modes:
            r += 1
        return r
    if select_modes == 'number':
        if type(n_modes) is not int:
            raise TypeError('The parameter n_modes is not an integer.')
        if not 1 <= n_modes <= n_cols:
            raise ValueError('The parameter n_modes is outside the [1-m] range.')
        return n_modes
    raise ValueError('The select_mode value is wrong.')


# ----------------------------------------------------------------------------
# a3  ROM.decomposition                            sparse_sensing.py:272-279
# ----------------------------------------------------------------------------
def decomposition(X0, select_modes='variance', n_modes=99):
    """Thin SVD, coefficient matrix, explained variance, truncation.

    :272 U,S,Vt = svd(X0, full_matrices=False); :273 A = (diag(S) Vt)^T;
    :274-275 exp_variance = 100 cumsum(S^2)/sum(S^2); :276 reduction.
    Returns (Ur, Ar, exp_variance[:r], S).
    """
    U, S, Vt = np.linalg.svd(X0, full_matrices=False)
    A = np.matmul(np.diag(S), Vt).T
    lam = S ** 2
    exp_variance = 100 * np.cumsum(lam) / np.sum(lam)
    r = select_rank(exp_variance, A.shape[1], select_modes, n_modes)
    return U[:, :r], A[:, :r], exp_variance[:r], S


# ----------------------------------------------------------------------------
# a5  ROM.fit                                      sparse_sensing.py:491-511
# ----------------------------------------------------------------------------
def fit(X, n_features, select_modes='variance', n_modes=99, axis_cnt=1, scale_type='std'):
    """scale_data -> decomposition -> Sigma_r / Vr (:504-508).

    Returns a dict with the attributes the reference object carries afterwards.
    """
    X_cnt, X_scl, X0 = scale_data(X, n_features, scale_type, axis_cnt)
    Ur, Ar, expv, S = decomposition(X0, select_modes, n_modes)
    r = Ar.shape[1]
    Sigma_r = np.zeros((r,))
    Vr = np.zeros_like(Ar)
    for i in range(r):
        Sigma_r[i] = np.linalg.norm(Ar[:, i])
        Vr[:, i] = Ar[:, i] / Sigma_r[i]
    return dict(X_cnt=X_cnt, X_scl=X_scl, X0=X0, Ur=Ur, Ar=Ar, r=r, Vr=Vr,
                Sigma_r=Sigma_r, exp_variance=expv, S=S)


# ----------------------------------------------------------------------------
# a6  SPR.optimal_placement('qr')                  sparse_sensing.py:735-743
# ----------------------------------------------------------------------------
def qr_pivots(Ur, mask=None):
    """Column-pivot order of the r x n matrix Ur^T (LAPACK dgeqp3 through SciPy).

    :737-738 rows outside ``mask`` are zeroed IN PLACE in the reference (the caller's
    Ur is modified); here a copy is zeroed and returned so the oracle stays pure.
    :739 ``la.qr(Ur.T, pivoting=True, mode='economic')``; only P[:r] is used (:740-743).
    Returns (P[:r] as int64, Ur_after_mask).
    """
    Ur = np.array(Ur, copy=True)
    if mask is not None:
        Ur[~mask, :] = 0
    _, _, P = sla.qr(Ur.T, pivoting=True, mode='economic')
    r = Ur.shape[1]
    return np.asarray(P[:r], dtype=np.int64), Ur


def gem_pivots(Ur, n_sensors, xyz, n_features, mask=None, d_min=0.0, noise=None, ridge=None):
    """SPR.gem (:586-698): greedy conditional-variance ("entropy") maximisation.

    Follows the reference statement by statement -- row variances with ddof=1 over the r entries (:621, :637),
    scaling coef = 2/sqrt(max variance) (:622-624), first pick = largest variance (:641), then per step the
    candidates filtered by the distance mask of the PREVIOUS pick (:655-657), S_aa = np.cov of the scaled
    picked rows (:660), its inverse (:662-668), and sigma2y_cond = sigma2y - S_ya S_aa^-1 S_ay (:670-678) with
    np.argmax's first-index tie-break (:681) -- except that the per-candidate Python loop of np.cov calls
    (:670) is evaluated for all candidates at once (same covariance entries: centred dot products / (r-1)).
    ``noise``: None -> the regularisation-free limit (what the device path computes);
               a callable k -> vector of length k to reproduce the reference's 1e-5*np.random.normal (:667).
    ``ridge``: None -> the reference's rule throughout.  A number d (the product uses 1e-5, the RMS of that noise):
               from the r-th pick on -- where S_aa of the r-1 picks already spans the centred space, every conditional
               variance is zero and the reference's choice is decided by its unseeded noise alone -- the noise is
               replaced by d on the whole diagonal (S_aa + d I), and rows already picked are no longer candidates.
               This is the documented deterministic stand-in of openmeasure_amd for n_sensors > r-1; parity with the
               reference is undefined there (its result changes from run to run).
    Returns the picked global rows (int64) and the relative lead of every pick over the runner-up."""
    Ur = np.asarray(Ur, dtype=np.float64)
    n, r = Ur.shape
    if mask is None:
        mask = np.ones((n,), dtype=bool)
    index_org = np.arange(n)
    sigma = np.var(Ur[mask], ddof=1, axis=1)
    coef = 1 / np.sqrt(sigma.max()) * 2
    Ur_msk = Ur[mask] * coef
    Ur_scl = Ur * coef
    xyz_msk = np.tile(xyz, (n_features, 1))[mask]
    index_msk = index_org[mask]
    picks, leads = [], []
    sigma_coef = np.var(Ur_msk, ddof=1, axis=1)

    def lead(v, i):
        rest = np.delete(v, i)
        return (v[i] - rest.max()) / v[i] if rest.size and v[i] > 0 else 0.0

    for s in range(n_sensors):
        if s == 0:
            temp = sigma_coef
        else:
            if ridge is not None and s >= r - 1:
                mask_d = mask_d & ~np.isin(index_msk, picks)
            Ur_msk, xyz_msk, index_msk = Ur_msk[mask_d], xyz_msk[mask_d], index_msk[mask_d]
            A = Ur_scl[picks, :]
            Ac = A - A.mean(axis=1, keepdims=True)
            S_aa = np.atleast_2d(Ac @ Ac.T / (r - 1))            # np.cov(A, ddof=1)
            if s == 1:
                S_inv = 1 / S_aa
            else:
                reg = np.zeros(s) if noise is None else np.asarray(noise(s))
                if ridge is not None and s >= r - 1:
                    reg = np.full(s, float(ridge))
                S_inv = np.linalg.inv(S_aa + np.diag(reg))
            Yc = Ur_msk - Ur_msk.mean(axis=1, keepdims=True)
            S_ya = Yc @ Ac.T / (r - 1)                            # rows: Sigma[-1, :-1] of every candidate
            sigma2y = np.sum(Yc * Yc, axis=1) / (r - 1)
            temp = sigma2y - np.sum((S_ya @ S_inv) * S_ya, axis=1)
        i_sensor = int(np.argmax(temp))
        picks.append(int(index_msk[i_sensor]))
        leads.append(lead(temp, i_sensor))
        d_sensor = np.linalg.norm(xyz_msk[i_sensor, :] - xyz_msk, axis=1)
        mask_d = d_sensor >= d_min
    return np.asarray(picks, dtype=np.int64), np.asarray(leads)


def one_hot_C(piv, n):
    """Dense s x n one-hot measurement matrix (:741-743)."""
    C = np.zeros((len(piv), n))
    for j, p in enumerate(piv):
        C[j, p] = 1
    return C


# ----------------------------------------------------------------------------
# a7  SPR.train                                    sparse_sensing.py:791-820
# ----------------------------------------------------------------------------
def train_theta(C, Ur, n, is_Theta=False):
    """Theta = C.dot(Ur) with the reference's two shape checks (:791-793, :801-803)."""
    if (C.shape[1] != n) and not is_Theta:
        raise ValueError('The number of columns of C does not match the number'
                         ' of rows of X.')
    Theta = C if is_Theta else C.dot(Ur)
    if Theta.shape[1] != Ur.shape[1]:
        raise ValueError('The number of columns of Theta does not match the number'
                         ' of columns of Ur.')
    return Theta


def theta_condition(Theta):
    """``train(cond=True)`` (:813-820): s1/s_last of Theta (square) or of pinv(Theta)."""
    if Theta.shape[0] == Theta.shape[1]:
        S = np.linalg.svd(Theta, compute_uv=False)
    else:
        S = np.linalg.svd(np.linalg.pinv(Theta), compute_uv=False)
    return S[0] / S[-1]


# ----------------------------------------------------------------------------
# a8  SPR.scale_vector                             sparse_sensing.py:571-584
# ----------------------------------------------------------------------------
def scale_vector(y, C, X_cnt, X_scl, n_points):
    """:573 cnt = C.X_cnt; :576 scl picked by the FEATURE COLUMN of y (col 2);
    :578-579 y0 = ((y - cnt)/scl, sigma/scl).  Returns (y0, cnt_vector, scl_vector)."""
    y0 = np.zeros((y.shape[0], 2))
    cnt_vector = C.dot(X_cnt[:, 0])
    scl_vector = X_scl[y[:, 2].astype('int') * n_points, 0]
    y0[:, 0] = (y[:, 0] - cnt_vector) / scl_vector
    y0[:, 1] = y[:, 1] / scl_vector
    return y0, cnt_vector, scl_vector


# ----------------------------------------------------------------------------
# a9  SPR.predict, method='OLS'                    sparse_sensing.py:844-901
# ----------------------------------------------------------------------------
def predict_ols(ys, Theta, C, X_cnt, X_scl, n_points):
    """(Weighted) least squares through the SVD pseudo-inverse, one vector at a time.

    :844-845 a bare array is one vector; :848-854 shape checks; :868-870 all-zero
    uncertainty -> W = I and zero sigma; :872-874 W = diag(1/y0[:,1]),
    sigma = |pinv(W Theta) y0[:,1]|; :877-878 a = pinv(W Theta) (W y0[:,0]).
    Returns (Ar (n_p,r), Ar_sigma (n_p,r)).
    """
    if isinstance(ys, np.ndarray):
        ys = [ys]
    for y in ys:
        if Theta.shape[0] != y.shape[0]:
            raise ValueError('The number of rows of Theta does not match the number'
                             ' of rows of y.')
        if y.shape[1] != 3:
            raise ValueError('The y array has the wrong number of columns. y has'
                             ' to have dimensions (s,3).')
    r = Theta.shape[1]
    Ar = np.zeros((len(ys), r))
    Ar_sigma = np.zeros((len(ys), r))
    for i, y in enumerate(ys):
        y0, _, _ = scale_vector(y, C, X_cnt, X_scl, n_points)
        if not np.any(y[:, 1]):
            W = np.eye(y.shape[0])
            ar_sigma = np.zeros((r,))
        else:
            W = np.diag(1 / y0[:, 1])
            ar_sigma = np.abs(np.dot(np.linalg.pinv(W @ Theta), y0[:, 1]))
        ar = np.dot(np.linalg.pinv(W @ Theta), W @ y0[:, 0])
        Ar[i, :] = ar
        Ar_sigma[i, :] = ar_sigma
    return Ar, Ar_sigma


# ----------------------------------------------------------------------------
# a10/a11  ROM.reconstruct + unscale_data          sparse_sensing.py:362-375, :235
# ----------------------------------------------------------------------------
def unscale(x0, X_cnt, X_scl):
    """:235 x = X_scl[:,0] * x0 + X_cnt[:,0] (the reference builds it as a cvxpy
    constant expression whose ``.value`` is exactly this multiply-then-add)."""
    return np.multiply(X_scl[:, 0], x0) + X_cnt[:, 0]


def unscale_sampled(x0, sampling, X_cnt, X_scl):
    """:233 x = (sampling @ X_scl[:,0]) * x0 + sampling @ X_cnt[:,0]."""
    return np.multiply(sampling @ X_scl[:, 0], x0) + sampling @ X_cnt[:, 0]


def reconstruct_sampled(Ar, Ur, X_cnt, X_scl, sampling):
    """:365-368 X_rec = sampling . Ur . Ar^T, un-scaled column by column with the sampled scale/centre.
    Returns (s, n_p)."""
    if Ar.ndim < 2:
        Ar = Ar[np.newaxis, :]
    X_rec = np.linalg.multi_dot([sampling, Ur, Ar.T])
    for i in range(X_rec.shape[1]):
        X_rec[:, i] = unscale_sampled(X_rec[:, i], sampling, X_cnt, X_scl)
    return X_rec


def reconstruct(Ar, Ur, X_cnt, X_scl):
    """:362-363 1-D -> (1,r); :371 X_rec = Ur @ Ar.T; :372-373 unscale column by column.
    Returns (n, n_p)."""
    if Ar.ndim < 2:
        Ar = Ar[np.newaxis, :]
    X_rec = Ur @ Ar.T
    for i in range(X_rec.shape[1]):
        X_rec[:, i] = unscale(X_rec[:, i], X_cnt, X_scl)
    return X_rec


# ----------------------------------------------------------------------------
# whole path, for the timed CPU baseline and end-to-end parity
# ----------------------------------------------------------------------------
def fit_place_train_predict_reconstruct(X, n_features, n_modes, y_fn):
    """fit('number') -> optimal_placement -> train -> predict -> reconstruct.

    ``y_fn(piv)`` returns the (s,3) measurement array for the chosen sensors.
    Returns dict(piv, Theta, Ar, Ar_sigma, X_rec, **fit state).
    """
    st = fit(X, n_features, 'number', n_modes)
    n = X.shape[0]
    piv, Ur_m = qr_pivots(st['Ur'])
    C = one_hot_C(piv, n)
    Theta = train_theta(C, Ur_m, n)
    y = y_fn(piv)
    Ar, Ar_sigma = predict_ols(y, Theta, C, st['X_cnt'], st['X_scl'], n // n_features)
    X_rec = reconstruct(Ar, Ur_m, st['X_cnt'], st['X_scl'])
    st.update(piv=piv, Theta=Theta, Ar_pred=Ar, Ar_sigma=Ar_sigma, X_rec=X_rec)
    return st


def fit_reconstruct_timed(X, n_features, n_modes):
    """The headline metric's CPU leg: fit('number', n_modes) + reconstruct(one vector).

    Uses row 0 of the fitted Ar as the coefficient vector (any r-vector costs the same).
    Returns (X_rec, state).
    """
    st = fit(X, n_features, 'number', n_modes)
    X_rec = reconstruct(st['Ar'][0, :], st['Ur'], st['X_cnt'], st['X_scl'])
    return X_rec, st
