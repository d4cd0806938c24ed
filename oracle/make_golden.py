#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference; the GPU box never runs it).
The reference module ``openmeasure/sparse_sensing.py`` imports cvxpy at module top
(:15) and cvxpy is not installed here.  The single cvxpy symbol executed on the SPR
path is ``cp.multiply(a, b) + c`` followed by ``.value`` (:233-238), i.e. an
element-wise multiply-then-add of float64 arrays.  This script registers an
in-memory module named ``cvxpy`` that offers exactly that (``multiply`` returning an
object with ``.value`` and ``+``), nothing else, then imports the reference file as
it lies on disk.  No reference source is copied, no bytecode is written.
Every X_rec fixture is additionally checked here against the plain NumPy expression
``X_scl*x0 + X_cnt`` so the stand-in cannot leak into the expected values.

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden')


class _ConstExpr:
    def __init__(self, v):
        self.value = v

    def __add__(self, o):
        return _ConstExpr(self.value + (o.value if isinstance(o, _ConstExpr) else o))

    __radd__ = __add__


def _import_reference():
    cp = types.ModuleType('cvxpy')
    cp.multiply = lambda a, b: _ConstExpr(np.multiply(a, b))
    sys.modules['cvxpy'] = cp
    sys.path.insert(0, '/root/reference/src')
    import openmeasure.sparse_sensing as sps
    return sps


def synth(n_points, n_features, m, k, rho, eps, seed):
    """Low-rank + noise snapshot matrix with a designed spectrum, then a per-feature
    affine map so centring and scaling are non-trivial."""
    rng = np.random.default_rng(seed)
    n = n_points * n_features
    L = rng.standard_normal((n, k))
    R = (rho ** np.arange(k))[:, None] * rng.standard_normal((k, m))
    X = L @ R + eps * rng.standard_normal((n, m))
    for f in range(n_features):
        X[f * n_points:(f + 1) * n_points] = (f + 1) * X[f * n_points:(f + 1) * n_points] + 10.0 * f
    return np.ascontiguousarray(X)


ONLY = set(sys.argv[1:])


def run_case(sps, name, X, n_features, select_modes, n_modes, seed, mask_frac=None, store_X0=False,
             scale_type='std', axis_cnt=1):
    if ONLY and name not in ONLY:                            # `make_golden.py name ...` regenerates just those cases
        return
    n, m = X.shape
    n_points = n // n_features
    rng = np.random.default_rng(seed + 7)
    xyz = rng.random((n_points, 3))
    spr = sps.SPR(X.copy(), n_features, xyz)
    spr.fit(scale_type=scale_type, axis_cnt=axis_cnt, select_modes=select_modes, n_modes=n_modes)
    out = dict(X=X, scale_type=np.array(scale_type), axis_cnt=np.int64(-1 if axis_cnt is None else axis_cnt), n_features=np.int64(n_features), select_modes=np.array(select_modes),
               n_modes=np.float64(n_modes), X_cnt=spr.X_cnt, X_scl=spr.X_scl,
               Ur=np.array(spr.Ur), Ar=np.array(spr.Ar), Vr=spr.Vr, Sigma_r=spr.Sigma_r,
               r=np.int64(spr.r))
    if store_X0:
        out['X0'] = spr.X0
    # singular values / explained variance as the reference computes them (:272-275)
    _, _, expv = spr.decomposition(spr.X0, select_modes, n_modes)
    out['exp_variance'] = expv
    out['S_full'] = np.linalg.svd(spr.X0, full_matrices=False)[1]   # same dgesdd job as :272

    mask = None
    if mask_frac is not None:
        mask = rng.random(n) < mask_frac
        out['mask'] = mask
    C = spr.optimal_placement(mask=mask)          # zeroes Ur rows in place when masked (:738)
    piv = np.argmax(C, axis=1).astype(np.int64)
    assert C.sum() == spr.r and (C.sum(axis=1) == 1).all()
    out['piv'] = piv
    out['C_shape'] = np.array(C.shape, dtype=np.int64)
    if mask is not None:                             # only differs from Ur when masked
        out['Ur_after_placement'] = np.array(spr.Ur)
    spr.train(C, cond=True)
    out['Theta'] = spr.Theta
    out['k'] = np.float64(spr.k)

    # measurement vectors: held-out states = random combinations of snapshots + noise
    s = spr.r
    ys = []
    for j in range(3):
        w = rng.standard_normal(m) / np.sqrt(m)
        xt = X @ w + X.mean(axis=1) * (1 - w.sum())
        y = np.zeros((s, 3))
        y[:, 0] = C @ xt
        y[:, 2] = piv // n_points
        if j == 1:                                   # weighted branch (:872-874)
            y[:, 1] = 0.01 * (1 + rng.random(s)) * np.abs(y[:, 0]).mean()
        ys.append(y)
    out['ys'] = np.stack(ys)
    y0 = spr.scale_vector(ys[1])
    out['y0_1'] = y0
    out['cnt_vector'] = spr.cnt_vector
    out['scl_vector'] = spr.scl_vector
    A1, S1 = spr.predict(ys[0])                      # single ndarray form (:844-845)
    A3, S3 = spr.predict(ys)                         # list form
    out['Ar_pred1'], out['Ar_sigma1'] = A1, S1
    out['Ar_pred3'], out['Ar_sigma3'] = A3, S3
    X1 = spr.reconstruct(A1[0])                      # 1-D coefficient vector (:362-363)
    X3 = spr.reconstruct(A3)
    chk = (spr.Ur @ A3.T) * spr.X_scl + spr.X_cnt    # stand-in independence check
    assert np.array_equal(chk, X3), 'cvxpy stand-in changed the arithmetic'
    out['X_rec1'], out['X_rec3'] = X1, X3
    # partial-field reconstruction through a sampling matrix (:365-368, :232-233): 7 rows, mixed one-hot / averaging
    S = np.zeros((7, n))
    S[np.arange(4), rng.integers(0, n, 4)] = 1.0
    for k in range(4, 7):
        cols = rng.integers(0, n, 5)
        S[k, cols] = rng.random(5)
    out['sampling'] = S
    out['X_rec3_sampled'] = spr.reconstruct(A3, sampling=S)
    out['unscale_sampled'] = spr.unscale_data(np.linspace(-1, 1, 7), sampling=S)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: n={n} m={m} r={spr.r} piv[:6]={piv[:6]} cond={spr.k:.3g} '
          f'-> {os.path.getsize(path)/1e6:.2f} MB')


def run_gem_case(sps, name, X, n_features, n_modes, n_sensors, seed, d_min=0.0, mask_frac=None, xyz_dim=3):
    """calc_type='gem' (:586-698).  The reference regularises with UNSEEDED noise (:667); the case is run under
    three np.random seeds and only kept if the picks agree, so the fixture pins the noise-free algorithm."""
    if ONLY and name not in ONLY:
        return
    n = X.shape[0]
    n_points = n // n_features
    rng = np.random.default_rng(seed + 7)
    xyz = rng.random((n_points, xyz_dim))
    mask = (rng.random(n) < mask_frac) if mask_frac is not None else None
    spr = sps.SPR(X.copy(), n_features, xyz)
    spr.fit(select_modes='number', n_modes=n_modes)
    picks = []
    for np_seed in (0, 1, 2):
        np.random.seed(np_seed)
        C = spr.optimal_placement(calc_type='gem', n_sensors=n_sensors, mask=mask, d_min=d_min)
        assert C.shape == (n_sensors, n)
        picks.append(np.argmax(C, axis=1))
    assert all(np.array_equal(picks[0], p) for p in picks[1:]), f'{name}: picks depend on the noise seed {picks}'
    out = dict(X=X, n_features=np.int64(n_features), n_modes=np.int64(n_modes), n_sensors=np.int64(n_sensors),
               xyz=xyz, d_min=np.float64(d_min), gem_piv=picks[0].astype(np.int64), Ur=spr.Ur.copy(), Ar=spr.Ar.copy(),
               C_shape=np.array(C.shape))
    if mask is not None:
        out['mask'] = mask
    # the placement feeds train -> predict -> reconstruct (docs/sparse_sensing_doc.ipynb cells 10-14): fewer sensors
    # than modes, so np.linalg.pinv (:873-878) returns the MINIMUM-NORM coefficients
    out.update(_predict_block(spr, C, picks[0], X, n_points, rng))
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: n={n} r={spr.r} gem picks={picks[0]} -> {os.path.getsize(path)/1e6:.2f} MB')


def _predict_block(spr, C, rows, X, n_points, rng, W_spread=None, drift_tol=1e-7):
    """train(C) -> predict(3 vectors, the middle one weighted) -> reconstruct through the reference, plus a stability
    check: pinv's rcond cut (1e-15 sigma_max) sits at rounding level, so a case is only kept when a 1e-13 relative
    perturbation of the measurements moves the coefficients by less than 1e-7 (i.e. no singular value straddles
    the cut).  rows: the global row each row of C samples (feature id of the measurement)."""
    m = X.shape[1]
    s = C.shape[0]
    spr.train(C)
    ys = []
    for j in range(3):
        w = rng.standard_normal(m) / np.sqrt(m)
        xt = X @ w + X.mean(axis=1) * (1 - w.sum())
        y = np.zeros((s, 3))
        y[:, 0] = C @ xt
        y[:, 2] = rows // n_points
        if j == 1:
            sig = 0.01 * (1 + rng.random(s)) * np.abs(y[:, 0]).mean()
            if W_spread is not None:                 # weights spread over several decades: cond(W Theta) grows with it
                sig = sig * W_spread ** rng.random(s)
            y[:, 1] = sig
        ys.append(y)
    A3, S3 = spr.predict(ys)
    ys_p = [y.copy() for y in ys]
    for y in ys_p:
        y[:, 0] *= 1 + 1e-13 * rng.standard_normal(s)
    A3p, _ = spr.predict(ys_p)
    drift = np.abs(A3p - A3).max() / np.abs(A3).max()
    assert drift < drift_tol, f'pinv result unstable under a 1e-13 perturbation ({drift:.2e}): not a usable fixture'
    X3 = spr.reconstruct(A3)
    sv = [np.linalg.svd((np.diag(1 / (y[:, 1] / spr.scl_vector)) if y[:, 1].any() else np.eye(s)) @ spr.Theta,
                        compute_uv=False) for y in ys]
    return dict(Theta=spr.Theta.copy(), ys=np.stack(ys), Ar_pred3=A3, Ar_sigma3=S3, X_rec3=X3, X_cnt=spr.X_cnt,
                X_scl=spr.X_scl, sv_WTheta=np.stack([np.pad(v, (0, max(0, spr.r - len(v)))) for v in sv]))


def run_pinv_case(sps, name, X, n_features, n_modes, seed, kind):
    """predict() on systems where np.linalg.pinv's SVD semantics matter (:873-878):
      under   -- the first r-2 rows of the QR placement: fewer sensors than modes, minimum-norm solution;
      dup     -- r sensors of which two are the same row: W Theta has rank r-1;
      zerocol -- Theta handed over through train(is_Theta=True)-free route is not possible (scale_vector needs C), so
                 the basis gets an exactly-zero column through fit(basis=...) instead: rank r-1 with an exact zero;
      illcond -- a basis whose column r-2 is column 0 plus 1e-7 of itself (fit(basis=...)), 3r random sensors: full
                 column rank, cond(W Theta) ~ 1e7-1e8, beyond the refined normal equations, well inside pinv's range."""
    if ONLY and name not in ONLY:
        return
    n, m = X.shape
    n_points = n // n_features
    rng = np.random.default_rng(seed + 7)
    spr = sps.SPR(X.copy(), n_features, None)
    spr.fit(select_modes='number', n_modes=n_modes)
    out = dict(X=X, n_features=np.int64(n_features), n_modes=np.int64(n_modes), kind=np.array(kind))
    r = spr.r
    spread = None
    if kind == 'zerocol':
        Ur, Ar = np.array(spr.Ur), np.array(spr.Ar)
        Ur[:, r - 2] = 0.0
        spr.fit(basis=(Ur, Ar))
    if kind == 'illcond':
        Ur, Ar = np.array(spr.Ur), np.array(spr.Ar)
        Ur[:, r - 2] = Ur[:, 0] + 1e-7 * Ur[:, r - 2]
        spr.fit(basis=(Ur, Ar))
    C = spr.optimal_placement() if kind in ('under', 'dup') else None
    if kind == 'under':
        C = C[:r - 2]
        rows = np.argmax(C, axis=1)
    elif kind == 'dup':
        C[r - 1] = C[0]
        rows = np.argmax(C, axis=1)
    elif kind == 'zerocol':
        rows = np.sort(rng.choice(n, r + 3, replace=False))
        C = np.zeros((r + 3, n)); C[np.arange(r + 3), rows] = 1.0
    elif kind == 'illcond':
        rows = np.sort(rng.choice(n, 3 * r, replace=False))
        C = np.zeros((3 * r, n)); C[np.arange(3 * r), rows] = 1.0
        spread = 1e2
    else:
        raise ValueError(kind)
    out['Ur'], out['Ar'] = np.array(spr.Ur), np.array(spr.Ar)
    out['C_rows'] = rows.astype(np.int64)
    # an ill-conditioned full-rank system amplifies the 1e-13 probe by its condition number: looser stability bar
    out.update(_predict_block(spr, C, rows, X, n_points, rng, W_spread=spread,
                              drift_tol=1e-3 if kind == 'illcond' else 1e-7))
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    sv = out['sv_WTheta']
    print(f'{name}: s={C.shape[0]} r={r} sv(W Theta) max/min per vector: '
          + ', '.join(f'{v.max():.2e}/{v[v > 0].min():.2e}' for v in sv))


def run_limits_case(sps, name, X, n_features, seed):
    """ROM.scale_limits (:173-210), the method GPR users call on the base class (tests/test_gpr_data.py:95): one
    ordinary pair of limits and one that trips the +-1000 clamps."""
    if ONLY and name not in ONLY:
        return
    rng = np.random.default_rng(seed)
    spr = sps.SPR(X.copy(), n_features, None)
    spr.fit(select_modes='number', n_modes=3)
    n_points = X.shape[0] // n_features
    lo = np.array([X[f * n_points:(f + 1) * n_points].min() for f in range(n_features)]) - rng.random(n_features)
    hi = np.array([X[f * n_points:(f + 1) * n_points].max() for f in range(n_features)]) + rng.random(n_features)
    out = dict(X=X, n_features=np.int64(n_features), lo=lo, hi=hi, lo_far=lo - 1e4 * spr.X_scl[::n_points, 0],
               hi_far=hi + 1e4 * spr.X_scl[::n_points, 0])
    l0 = spr.scale_limits([lo, hi])
    l1 = spr.scale_limits([out['lo_far'], out['hi_far']])
    out.update(lim0_lo=l0[0], lim0_hi=l0[1], lim1_lo=l1[0], lim1_hi=l1[1])
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: limits0 range [{l0[0].min():.3g}, {l0[1].max():.3g}] clamped [{l1[0].min():.3g}, {l1[1].max():.3g}]')


def main():
    sps = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    # G1: the reference tests' own shape (tests/test_rom.py:9-12), seeded
    X = np.random.default_rng(101).random((20, 5))
    run_case(sps, 'g1_full', X, 2, 'variance', 100, 101, store_X0=True)
    run_case(sps, 'g1_num4', X, 2, 'number', 4, 102, store_X0=True)
    # G2: 500 cells x 3 features x 12 snapshots, r in {4, 12}, default 99 % variance
    X = synth(500, 3, 12, 12, 0.7, 1e-3, 202)
    run_case(sps, 'g2_num4', X, 3, 'number', 4, 203, store_X0=True)
    run_case(sps, 'g2_full', X, 3, 'variance', 100, 204)
    run_case(sps, 'g2_var99', X, 3, 'variance', 99, 205)
    run_case(sps, 'g2_num4_mask', X, 3, 'number', 4, 206, mask_frac=0.5)
    # G3: 2000 x 4 x 24 with a designed spectrum (sigma_1/sigma_16 ~ 1e3), r = s = 8, 16
    X = synth(2000, 4, 24, 24, 10 ** (-3 / 15), 1e-4, 303)
    run_case(sps, 'g3_num8', X, 4, 'number', 8, 304)
    run_case(sps, 'g3_num16', X, 4, 'number', 16, 305)
    # G7 (round 6): 9 features like BASELINE configs 3 / 4 -- 1000 cells x 9 x 16 snapshots, r = 6.  Eight equal row blocks
    # are 1125 rows each: every interior block boundary falls INSIDE a feature (multiples of 1125 against multiples of 1000)
    X = synth(1000, 9, 16, 16, 0.75, 1e-3, 1707)
    run_case(sps, 'g7_f9_num6', X, 9, 'number', 6, 1708)
    # odd m / odd r / ragged: 333 cells x 3 x 7 snapshots, r = 5
    X = synth(333, 3, 7, 7, 0.6, 1e-3, 404)
    run_case(sps, 'g4_num5', X, 3, 'number', 5, 405)
    # the other per-feature scalings of ROM.scale_data (:117-161) on the G2 matrix (positive data so
    # that 'level' / 'poisson' / 'vast' are well defined)
    X = synth(500, 3, 12, 12, 0.7, 1e-3, 202) * 0.05 + 5.0
    for k, st in enumerate(['none', 'pareto', 'vast', 'level', 'variance', 'poisson', 'l2-norm', 'range', 'max', 'median']):
        run_case(sps, 'g5_' + st.replace('-', ''), X, 3, 'number', 4, 500 + k, scale_type=st)
    # scalar centring per feature (axis_cnt=None, tests/test_rom.py:23-29)
    run_case(sps, 'g6_axisnone', X, 3, 'number', 4, 600, axis_cnt=None)
    run_case(sps, 'g6_axisnone_pareto', X, 3, 'number', 5, 601, axis_cnt=None, scale_type='pareto')
    run_limits_case(sps, 'lim_small', synth(40, 3, 8, 8, 0.7, 1e-3, 808), 3, 809)
    # GEM placement: distance exclusion, search mask, 2-D coordinates, the full r-1 sensors
    X = synth(300, 3, 12, 12, 0.7, 1e-3, 707)
    run_gem_case(sps, 'gem_dmin', X, 3, 8, 6, 701, d_min=0.15)
    run_gem_case(sps, 'gem_mask', X, 3, 8, 5, 702, mask_frac=0.5)
    run_gem_case(sps, 'gem_xz_full', X, 3, 6, 5, 703, d_min=0.05, xyz_dim=2)
    # predict() where pinv's minimum-norm / rank-revealing semantics matter (:873-878)
    X = synth(400, 3, 16, 16, 0.7, 1e-3, 909)
    for k, kind in enumerate(['under', 'dup', 'zerocol', 'illcond']):
        run_pinv_case(sps, 'pinv_' + kind, X, 3, 8, 910 + k, kind)
    # float32 snapshot matrices: X_cnt / X_scl are float64 (np.zeros, :106-107), so X0 = (X - X_cnt)/X_scl (:169), U (:272),
    # Ur and Ar come out float64 -- the means and the std themselves are formed by NumPy in float32 (np.average / np.std of
    # a float32 block).  The fixtures store X as float32; consumers read Ur.dtype from the stored array.
    X = synth(500, 3, 12, 12, 0.7, 1e-3, 202).astype(np.float32)
    run_case(sps, 'f32_g2_num4', X, 3, 'number', 4, 1001)
    X = synth(2000, 4, 24, 24, 10 ** (-3 / 15), 1e-4, 303).astype(np.float32)
    run_case(sps, 'f32_g3_num8', X, 4, 'number', 8, 1002)
    # conditioning of the Gram route (hard part 1): designed spectra with sigma_1/sigma_r = 1e5 and 1e7
    for tag, decades in (('1e5', 5), ('1e7', 7)):
        X = synth(1500, 3, 20, 20, 10 ** (-decades / 9), 1e-11, 920 + decades)
        run_case(sps, 'cond_' + tag, X, 3, 'number', 10, 930 + decades)


if __name__ == '__main__':
    main()
