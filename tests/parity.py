"""Parity checks of openmeasure_amd.SPR against a golden fixture -- shared by the CPU
tests (NumPy test double engine: host logic only) and the GPU tests (HIP engine)."""
import numpy as np

from openmeasure_amd.sparse_sensing import SPR

REL_FRO = 1e-6          # north_star: reconstructed fields within 1e-6 relative Frobenius


def rel_fro(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def align_signs(A, B):
    """column signs that best map A onto B"""
    s = np.sign(np.sum(A * B, axis=0))
    s[s == 0] = 1.0
    return s


def well_defined_rank(g):
    """modes whose singular value is far enough above the Gram route's noise floor"""
    S = g['S_full']
    return int(np.sum(S[:g['r']] > 1e-6 * S[0]))


def run_fixture(g, engine, check_pivots=True):
    """fit -> optimal_placement -> train -> predict -> reconstruct on fixture g; returns the model."""
    X = g['X'].copy()
    n, m = X.shape
    F = g['n_features']
    # float32 X (fixtures f32_*): the reference forms the row means and the feature std in float32 (np.average / np.std
    # of a float32 block) and everything after that in float64 (:106-107, :169); the device forms them in float64 from
    # the same float32 values, so centring-dependent quantities agree to float32 rounding of the means, the
    # reconstructed FIELD still to 1e-6, and the basis dtype must be the reference's float64
    lo = X.dtype == np.float32
    t = (lambda tight, loose: loose if lo else tight)
    spr = SPR(X, F, None, engine=engine)
    spr.fit(scale_type=g.get('scale_type', 'std'), axis_cnt=g.get('axis_cnt', 1), select_modes=g['select_modes'],
            n_modes=g['n_modes'])
    r = g['r']
    assert spr.r == r
    # a2: centring / scaling
    np.testing.assert_allclose(spr.X_cnt, g['X_cnt'], rtol=t(1e-13, 1e-6), atol=t(1e-13, 1e-6) * np.abs(g['X_cnt']).max())
    np.testing.assert_allclose(spr.X_scl, g['X_scl'], rtol=t(1e-12, 1e-6))
    assert spr.X_cnt.dtype == np.float64 and spr.X_scl.dtype == np.float64
    assert spr.X_cnt.shape == (n, 1) and spr.X_scl.shape == (n, 1)
    # a3/a5: spectrum and basis (up to column sign; only modes above the noise floor)
    rw = well_defined_rank(g)
    S0 = g['S_full'][0]
    np.testing.assert_allclose(spr.Sigma_r[:rw], g['Sigma_r'][:rw], rtol=t(1e-8, 1e-5), atol=1e-9 * S0)
    np.testing.assert_allclose(spr.exp_variance_[:rw], g['exp_variance'][:rw], rtol=t(1e-9, 1e-6))
    sg = align_signs(spr.Ar[:, :rw], g['Ar'][:, :rw])
    np.testing.assert_allclose(spr.Ar[:, :rw] * sg, g['Ar'][:, :rw], rtol=0, atol=t(1e-8, 1e-4) * S0)
    np.testing.assert_allclose(spr.Vr[:, :rw] * sg, g['Vr'][:, :rw], rtol=0, atol=t(1e-7, 1e-4))
    np.testing.assert_allclose(spr.Ur[:, :rw] * sg, g['Ur'][:, :rw], rtol=0, atol=t(1e-8, 1e-4))
    assert spr.Ur.shape == (n, r) and spr.Ar.shape == (m, r)
    assert spr.Ur.dtype == g['Ur'].dtype == np.float64 and spr.Ar.dtype == np.float64   # also for a float32 X (:169, :272)
    # a6: sensors -- exact, ordered; only meaningful when every retained mode is well defined
    mask = g.get('mask')
    C = spr.optimal_placement(mask=mask)
    assert C.shape == tuple(g['C_shape'])
    if rw == r and check_pivots:
        np.testing.assert_array_equal(spr.sensors_, g['piv'])
        np.testing.assert_array_equal(np.argmax(np.asarray(C), axis=1), g['piv'])
        assert spr.pivot_gap_.min() > 1e-9
    if mask is not None:
        assert not spr.Ur[~mask].any()                       # :737-738 zeroes the basis rows in place
    # from here on compare on the reference's own sensors (independent of pivot parity)
    piv = g['piv']
    Cg = np.zeros((len(piv), n))
    Cg[np.arange(len(piv)), piv] = 1
    spr.train(Cg, cond=True)
    sg_r = np.ones(r)
    sg_r[:rw] = sg
    if rw == r:
        np.testing.assert_allclose(spr.Theta * sg_r, g['Theta'], rtol=0, atol=t(1e-8, 1e-4))
        assert spr.k == np.float64(spr.k) and abs(spr.k - float(g['k'])) <= 1e-6 * float(g['k'])
    # a8
    y0 = spr.scale_vector(g['ys'][1])
    np.testing.assert_allclose(y0, g['y0_1'], rtol=t(1e-11, 1e-4), atol=t(1e-12, 1e-5))
    np.testing.assert_allclose(spr.cnt_vector, g['cnt_vector'], rtol=t(1e-13, 1e-6), atol=t(1e-13, 1e-6))
    np.testing.assert_allclose(spr.scl_vector, g['scl_vector'], rtol=t(1e-12, 1e-6))
    # a9 + a10: coefficients (sign-aligned) and the reconstructed fields
    A1, S1 = spr.predict(g['ys'][0])
    A3, S3 = spr.predict(list(g['ys']))
    assert A1.shape == (1, r) and A3.shape == (3, r) and S3.shape == (3, r)
    if rw == r:
        scale = np.abs(g['Ar_pred3']).max()
        np.testing.assert_allclose(A3 * sg_r, g['Ar_pred3'], rtol=0, atol=t(1e-7, 1e-4) * scale)
        np.testing.assert_allclose(S3, g['Ar_sigma3'], rtol=t(1e-6, 1e-3), atol=1e-9 * np.abs(g['Ar_sigma3']).max())
        np.testing.assert_allclose(A1 * sg_r, g['Ar_pred1'], rtol=0, atol=t(1e-7, 1e-4) * scale)
    assert not S1.any() and not S3[0].any() and S3[1].any()
    X1 = spr.reconstruct(A1[0])
    X3 = spr.reconstruct(A3)
    assert X1.shape == (n, 1) and X3.shape == (n, 3)
    if rw == r:
        assert rel_fro(X1, g['X_rec1']) <= REL_FRO
        assert rel_fro(X3, g['X_rec3']) <= REL_FRO
    # partial-field reconstruction / un-scaling through a sampling matrix (:365-368, :232-233)
    S = g['sampling']
    if rw == r:
        Xs = spr.reconstruct(g['Ar_pred3'] * sg_r, sampling=S)
        assert Xs.shape == g['X_rec3_sampled'].shape
        np.testing.assert_allclose(Xs, g['X_rec3_sampled'], rtol=t(1e-9, 1e-6), atol=t(1e-9, 1e-6) * np.abs(g['X_rec3_sampled']).max())
    np.testing.assert_allclose(spr.unscale_data(np.linspace(-1, 1, 7), sampling=S), g['unscale_sampled'], rtol=t(1e-11, 1e-6),
                               atol=t(1e-12, 1e-6) * np.abs(g['unscale_sampled']).max())
    import scipy.sparse as sps_
    np.testing.assert_allclose(spr.unscale_data(np.linspace(-1, 1, 7), sampling=sps_.csr_matrix(S)), g['unscale_sampled'],
                               rtol=t(1e-11, 1e-6), atol=t(1e-12, 1e-6) * np.abs(g['unscale_sampled']).max())
    return spr


class _GprLike:
    """Mixin reproducing the ROM calls of the reference's GPR.fit (gpr.py:379-402) and GPR.predict's
    reconstruct -- the GP itself (gpytorch) is outside the device path."""

    def fit_like_gpr(self, scale_type, axis_cnt, select_modes, n_modes):
        self.X0 = self.scale_data(scale_type, axis_cnt)
        Ur, Ar, _ = self.decomposition(self.X0, select_modes, n_modes)
        self.Ur = Ur
        self.Ar = Ar
        self.r = Ar.shape[1]
        self.Sigma_r = np.linalg.norm(Ar, axis=0)
        with np.errstate(invalid='ignore', divide='ignore'):   # the null mode of a full-rank row-centred fit
            self.Vr = Ar / self.Sigma_r


def run_gpr_style(g, engine, foreign_basis=False):
    """scale_data -> decomposition -> attribute assignment -> reconstruct, as a ROM subclass does it."""
    from openmeasure_amd.sparse_sensing import ROM

    class GprLike(_GprLike, ROM):
        pass

    X = g['X'].copy()
    rom = GprLike(X, g['n_features'], None, engine=engine)
    rom.fit_like_gpr(g.get('scale_type', 'std'), g.get('axis_cnt', 1), g['select_modes'], g['n_modes'])
    assert rom.r == g['r']
    X0_ref = g['X0'] if 'X0' in g else (g['X'] - g['X_cnt']) / g['X_scl']
    lo = g['X'].dtype == np.float32                            # see run_fixture: float32 means in the reference
    np.testing.assert_allclose(rom.X0, X0_ref, rtol=0, atol=(1e-5 if lo else 1e-12) * np.abs(X0_ref).max())
    assert rom.X0.dtype == np.float64
    rw = well_defined_rank(g)
    sg = align_signs(rom.Ar[:, :rw], g['Ar'][:, :rw])
    np.testing.assert_allclose(rom.Ur[:, :rw] * sg, g['Ur'][:, :rw], rtol=0, atol=1e-4 if lo else 1e-8)
    if foreign_basis:                                         # a basis that did not come from decomposition()
        rom.Ur = g['Ur'].copy()
        rom.Ar = g['Ar'].copy()
        Ar_pred = g['Ar'][:3]
    else:
        Ar_pred = rom.Ar[:3]
    X_rec = rom.reconstruct(Ar_pred)                          # gpr.py predict -> ROM.reconstruct
    want = g['X'][:, :3]
    # the first three training snapshots are reproduced up to the truncation error of the fixture's own basis
    ref = (g['Ur'] @ g['Ar'][:3].T) * g['X_scl'] + g['X_cnt']
    assert rel_fro(X_rec, ref) < REL_FRO
    assert X_rec.shape == want.shape
    return rom


def run_gem_fixture(g, engine):
    """fit -> optimal_placement(calc_type='gem') on a GEM fixture: picks exact and ordered.
    The variance of a row of Ur over its r entries changes when a column of Ur changes sign, so GEM's picks depend
    on LAPACK's (arbitrary) singular-vector signs: the fitted basis is checked against the fixture up to sign, then
    the placement runs on the reference's own basis through fit(basis=...) (:493-497)."""
    X = g['X'].copy()
    spr = SPR(X, g['n_features'], g['xyz'], engine=engine)
    spr.fit(select_modes='number', n_modes=g['n_modes'])
    sg = align_signs(spr.Ar, g['Ar'])
    np.testing.assert_allclose(spr.Ur * sg, g['Ur'], rtol=0, atol=1e-8)
    spr.fit(basis=(g['Ur'].copy(), g['Ar'].copy()))
    C = spr.optimal_placement(calc_type='gem', n_sensors=g['n_sensors'], mask=g.get('mask'), d_min=g['d_min'])
    assert C.shape == tuple(g['C_shape'])
    np.testing.assert_array_equal(spr.sensors_, g['gem_piv'])
    np.testing.assert_array_equal(np.argmax(np.asarray(C), axis=1), g['gem_piv'])
    spr.train(C)                                              # the placement feeds train/predict like the QR one
    assert spr.Theta.shape == (g['n_sensors'], spr.r)
    check_predict_block(spr, g)
    return spr


def check_predict_block(spr, g, coef_tol=1e-7, sigma_rtol=1e-6):
    """predict -> reconstruct against the reference's stored results on a model whose basis is the fixture's own
    (no sign ambiguity).  Fewer sensors than modes / rank-deficient W Theta: the reference's pinv (:873-878) returns
    the minimum-norm coefficients and so must the device path -- never a LinAlgError."""
    np.testing.assert_allclose(spr.Theta, g['Theta'], rtol=0, atol=1e-9 * np.abs(g['Theta']).max())
    ys = list(g['ys'])
    A3, S3 = spr.predict(ys)
    r = spr.r
    assert A3.shape == (3, r) and S3.shape == (3, r)
    scale = np.abs(g['Ar_pred3']).max()
    np.testing.assert_allclose(A3, g['Ar_pred3'], rtol=0, atol=coef_tol * scale)
    np.testing.assert_allclose(S3, g['Ar_sigma3'], rtol=sigma_rtol, atol=coef_tol * np.abs(g['Ar_sigma3']).max())
    assert not S3[0].any() and S3[1].any()
    A1, _ = spr.predict(ys[2])                                # single-array form (:844-845)
    np.testing.assert_allclose(A1[0], g['Ar_pred3'][2], rtol=0, atol=coef_tol * scale)
    X3 = spr.reconstruct(A3)
    assert X3.shape == g['X_rec3'].shape
    assert rel_fro(X3, g['X_rec3']) <= REL_FRO
    return A3, S3


def run_pinv_fixture(g, engine):
    """fit(basis=the fixture's) -> train(C) -> predict -> reconstruct where np.linalg.pinv's SVD semantics decide the
    answer: 'under' (s = r-2 sensors), 'dup' (a sensor twice: rank r-1), 'zerocol' (an exactly-zero basis column),
    'illcond' (two nearly collinear basis columns: cond(W Theta) ~ 1e7-1e8, full rank)."""
    X = g['X'].copy()
    n = X.shape[0]
    spr = SPR(X, g['n_features'], None, engine=engine)
    spr.fit(select_modes='number', n_modes=g['n_modes'])      # statistics (X_cnt, X_scl) from the data ...
    np.testing.assert_allclose(spr.X_cnt, g['X_cnt'], rtol=1e-13, atol=1e-13 * np.abs(g['X_cnt']).max())
    spr.fit(basis=(g['Ur'].copy(), g['Ar'].copy()))           # ... the basis exactly the reference's (:493-497)
    rows = g['C_rows']
    C = np.zeros((len(rows), n))
    C[np.arange(len(rows)), rows] = 1.0
    spr.train(C)
    # ill-conditioned but full rank: the coefficients themselves are only defined to cond * eps (the two nearly
    # collinear modes trade off against each other), the reconstructed field is not affected
    loose = g['kind'] == 'illcond'
    check_predict_block(spr, g, coef_tol=2e-4 if loose else 1e-7, sigma_rtol=1e-3 if loose else 1e-6)
    s_, r = spr.Theta.shape
    if g['kind'] != 'illcond' or True:
        assert spr.solve_path_ == 'pinv'                      # none of these is a case for the normal equations
    if g['kind'] in ('dup', 'zerocol'):
        assert (spr.solve_rank_ == r - 1).all()
    if g['kind'] == 'under':
        assert (spr.solve_rank_ == s_).all()
    return spr


def run_f32_storage(engine, n_points, F, m, r, seed, synth):
    """float32 snapshot matrix.  Default semantics (host float32 ndarray): stored as float32 in HBM, f64 arithmetic, FLOAT64
    basis like the reference (:106-107, :169, :272), sensors = the reference's.  Storage option
    DeviceMatrix(basis='f32') (BASELINE config 5): the basis is rounded to float32 once; everything is compared with the
    oracle run in f64 on the same f32-rounded values -- statistics and spectrum to f64 accuracy, the basis to f32 rounding,
    the sensors against BOTH the reference's choice (pivots of the oracle's f64 basis) and dgeqp3 on the stored basis
    widened to f64, with the pivot gaps far above f32 rounding; the field within 1e-6."""
    from oracle import spr_oracle as orc
    from openmeasure_amd.sparse_sensing import DeviceMatrix
    X32 = synth(n_points, F, m, min(m, 2 * r), 0.8, 1e-3, seed).astype(np.float32)
    Xw = X32.astype(np.float64)
    n = n_points * F
    st = orc.fit(Xw, F, 'number', r)
    piv_ref, _ = orc.qr_pivots(st['Ur'])
    # default: float64 basis, the reference's sensors
    dflt = SPR(X32, F, None, engine=engine)
    dflt.fit(select_modes='number', n_modes=r)
    assert dflt.Ur.dtype == np.float64 and dflt._d['X'].dtype == engine.torch.float32
    sgd = align_signs(dflt.Ar, st['Ar'])
    np.testing.assert_allclose(dflt.Ur * sgd, st['Ur'], rtol=0, atol=1e-8 * max(1.0, np.abs(st['Ur']).max()))
    dflt.optimal_placement()
    np.testing.assert_array_equal(dflt.sensors_, piv_ref)
    # storage option: float32 basis
    Xdev = engine.to_device(X32, dtype=engine.torch.float32)
    spr = SPR(DeviceMatrix(Xdev, basis='f32'), F, None, engine=engine)
    spr.fit(select_modes='number', n_modes=r)
    assert spr.Ur.dtype == np.float32 and spr.Ur.shape == (n, r)
    np.testing.assert_allclose(spr.X_cnt, st['X_cnt'], rtol=1e-12, atol=1e-12 * np.abs(st['X_cnt']).max())
    np.testing.assert_allclose(spr.X_scl, st['X_scl'], rtol=1e-11)
    np.testing.assert_allclose(spr.Sigma_r, st['Sigma_r'], rtol=1e-8)
    sg = align_signs(spr.Ar, st['Ar'])
    scale_u = np.abs(st['Ur']).max()
    # the stored basis is the f64 basis of the same path rounded to f32 (twice for m > 256: two column slices);
    # against the oracle's SVD route only up to the conditioning of the noise-level modes (clustered singular values)
    ref64 = SPR(Xw, F, None, engine=engine)
    ref64.fit(select_modes='number', n_modes=r)
    s64 = align_signs(ref64.Ar, spr.Ar)
    np.testing.assert_allclose(spr.Ur, ref64.Ur * s64, rtol=0, atol=2e-7 * scale_u)
    z = np.random.default_rng(seed).standard_normal(n)
    U32 = spr.Ur.astype(np.float64)
    assert rel_fro(U32 @ (U32.T @ z), st['Ur'] @ (st['Ur'].T @ z)) < 1e-5      # same subspace as the SVD route
    C = spr.optimal_placement()
    want, _ = orc.qr_pivots(spr.Ur.astype(np.float64))
    np.testing.assert_array_equal(spr.sensors_, want)
    # ... and they are the reference's sensors (pivots of the f64 basis): every pick led its runner-up by far more than
    # the f32 rounding of the stored basis (6e-8 relative per entry)
    np.testing.assert_array_equal(spr.sensors_, piv_ref)
    assert spr.pivot_gap_.min() > 1e-5, spr.pivot_gap_.min()
    spr.train(C)
    Cd = np.zeros((r, n)); Cd[np.arange(r), spr.sensors_] = 1.0
    ys = []
    for j in (0, 1):
        y = np.zeros((r, 3)); y[:, 0] = Xw[spr.sensors_, j]; y[:, 2] = spr.sensors_ // n_points
        if j == 1:
            y[:, 1] = 0.01 * (1.0 + np.arange(r) % 3)
        ys.append(y)
    Ar, Ar_sigma = spr.predict(ys)
    Ur_ref = st['Ur'] * sg                                    # oracle basis in the device's sign convention
    Theta_ref = orc.train_theta(Cd, Ur_ref, n)
    A_ref, S_ref = orc.predict_ols(ys, Theta_ref, Cd, st['X_cnt'], st['X_scl'], n_points)
    X_ref = orc.reconstruct(A_ref, Ur_ref, st['X_cnt'], st['X_scl'])
    X_rec = spr.reconstruct(Ar)
    assert X_rec.shape == (n, 2)
    assert rel_fro(X_rec, X_ref) <= REL_FRO
    # partial-field reconstruction through a sampling matrix (:365-368) on the f32-stored basis
    rng = np.random.default_rng(seed + 1)
    S = np.zeros((6, n))
    S[np.arange(3), rng.integers(0, n, 3)] = 1.0
    for k in range(3, 6):
        S[k, rng.integers(0, n, 4)] = rng.random(4)
    Xs = spr.reconstruct(Ar, sampling=S)
    Xs_ref = orc.reconstruct_sampled(A_ref, Ur_ref, st['X_cnt'], st['X_scl'], S)
    assert Xs.shape == Xs_ref.shape and rel_fro(Xs, Xs_ref) <= REL_FRO
    # masked placement zeroes the stored rows in place (:737-738) and pivots what is left
    mask = rng.random(n) < 0.5
    U_before = spr.Ur.astype(np.float64)
    spr.optimal_placement(mask=mask)
    want_m, U_masked = orc.qr_pivots(U_before, mask)
    np.testing.assert_array_equal(spr.sensors_, want_m)
    np.testing.assert_array_equal(spr.Ur.astype(np.float64), U_masked)
    return spr


def run_conditioning_guard(engine, decades, synth, f32=False, shape=(1500, 3, 20, 10)):
    """Gram route on a designed spectrum sigma_1/sigma_r ~ 10^decades: fit() must either return the reference's
    sensors exactly (refinement pass of _refine_spectrum) or refuse with LinAlgError -- never degrade silently."""
    from oracle import spr_oracle as orc
    n_points, F, m, r = shape
    X = synth(n_points, F, m, min(m, 2 * r), 10 ** (-decades / (r - 1)), 1e-16, 77 + decades)
    if f32:                                                   # f32 STORAGE: the problem is the rounded data, solved in f64
        X32 = X.astype(np.float32)
        X = X32.astype(np.float64)
    st = orc.fit(X, F, 'number', r)
    piv, _ = orc.qr_pivots(st['Ur'])
    kappa = st['Sigma_r'][0] / st['Sigma_r'][-1]
    if f32:
        from openmeasure_amd.sparse_sensing import DeviceMatrix
        spr = SPR(DeviceMatrix(engine.to_device(X32, dtype=engine.torch.float32), basis='f32'), F, None, engine=engine)
    else:
        spr = SPR(X, F, None, engine=engine)
    try:
        spr.fit(select_modes='number', n_modes=r)
    except np.linalg.LinAlgError as e:
        assert kappa > 1e8 and 'refinement' in str(e)
        return None
    assert (spr.gram_refine_passes_ > 0) == (kappa > 1e4)
    if f32:
        # the basis is stored in f32: the sensors are dgeqp3's pivots on the STORED basis; it agrees with the f64 basis of
        # the same data to f32 rounding
        sgn = align_signs(spr.Ar, st['Ar'])
        np.testing.assert_allclose(spr.Ur.astype(np.float64) * sgn, st['Ur'], rtol=0, atol=3e-7 * np.abs(st['Ur']).max())
        spr.optimal_placement()
        want, _ = orc.qr_pivots(spr.Ur.astype(np.float64))
        np.testing.assert_array_equal(spr.sensors_, want)
        np.testing.assert_allclose(spr.Sigma_r, st['Sigma_r'], rtol=max(1e-9, 100 * np.finfo(float).eps * kappa))
        return spr
    spr.optimal_placement()
    np.testing.assert_array_equal(spr.sensors_, piv)
    eps_k = np.finfo(float).eps * kappa
    np.testing.assert_allclose(spr.Sigma_r, st['Sigma_r'], rtol=max(1e-9, 100 * eps_k))
    sg = align_signs(spr.Ar, st['Ar'])
    np.testing.assert_allclose(spr.Ur * sg, st['Ur'], rtol=0, atol=max(1e-9, 1e3 * eps_k) * np.abs(st['Ur']).max())
    return spr


def run_gem_beyond_rank(engine, n_points, F, r, n_sensors, d_min, masked, seed, xyz_dim=3):
    """calc_type='gem' with more sensors than r-1: the first r-1 picks are the noise-free rule (pinned by the
    reference fixtures), the rest follow the documented ridge stand-in for the reference's unseeded noise --
    compared with the oracle's literal covariance formulas (gem_pivots(ridge=1e-5)): exact and ordered."""
    from oracle import spr_oracle as orc
    rng = np.random.default_rng(seed)
    n = n_points * F
    Ur, _ = np.linalg.qr(rng.standard_normal((n, r)) * (1.0 + 3.0 * rng.random((n, 1))))
    xyz = rng.random((n_points, xyz_dim))
    mask = (rng.random(n) < 0.6) if masked else None
    want, lead = orc.gem_pivots(Ur, n_sensors, xyz, F, mask, d_min, ridge=1e-5)
    assert lead[:r - 1].min() > 1e-6 and lead[r - 1:].min() > 1e-6   # no near-ties: the comparison is meaningful
    spr = SPR(rng.standard_normal((n, 4)), F, xyz, engine=engine)
    spr.fit(basis=(Ur.copy(), np.eye(4, r)))
    C = spr.optimal_placement(calc_type='gem', n_sensors=n_sensors, mask=mask, d_min=d_min)
    assert C.shape == (n_sensors, n)
    np.testing.assert_array_equal(spr.sensors_, want)
    assert len(set(spr.sensors_.tolist())) == n_sensors
    first, _ = orc.gem_pivots(Ur, r - 1, xyz, F, mask, d_min)          # the reference's own (noise-free) rule
    np.testing.assert_array_equal(spr.sensors_[:r - 1], first)
    return spr



def run_documented_idioms(g, engine, monkeypatch):
    """The reference's documented use of the measurement matrix (README.md:160-184 and INTEGRATION.md) run twice: on
    the dense ndarray optimal_placement returns while it is small, and on the OneHotRows it returns above the dense
    limit (46 GB at BASELINE config 3) -- every line must behave the same."""
    import openmeasure_amd.rom as mod
    X = g['X'].astype(np.float64)
    n, m = X.shape
    F = g['n_features']
    n_cells = n // F
    r = g['r']
    x_test = X @ (np.arange(1, m + 1) / (m * (m + 1) / 2))         # a state in the span of the data
    outs = []
    for limit in (mod._DENSE_C_LIMIT, 0):
        monkeypatch.setattr(mod, '_DENSE_C_LIMIT', limit)
        spr = SPR(X.copy(), F, None, engine=engine)
        spr.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])
        C_qr = spr.optimal_placement()
        assert isinstance(C_qr, np.ndarray if limit else mod.OneHotRows) and C_qr.shape == (r, n)
        xz_sensors = np.zeros((r, 4))
        for i in range(r):                                         # README.md:166-169
            index = np.argmax(C_qr[i, :])
            xz_sensors[i, 2] = index // n_cells
            xz_sensors[i, 3] = index % n_cells
        y_qr = np.ones((r, 3))
        y_qr[:, 0] = C_qr @ x_test                                  # README.md:176
        y_qr[:, 1] = 0.0
        for i in range(r):                                          # README.md:178-179
            y_qr[i, 2] = np.argmax(C_qr[i, :]) // n_cells
        y2 = np.zeros((r, 3)); y2[:, 0] = C_qr.dot(x_test)
        y2[:, 2] = np.argmax(C_qr, axis=1) // spr.n_points          # INTEGRATION.md (vectorised form of the loop)
        np.testing.assert_array_equal(y2, y_qr)
        assert np.asarray(C_qr).shape == (r, n) and np.asarray(C_qr).sum() == r
        spr.train(C_qr)                                             # README.md:182
        ap, sigmap = spr.predict(y_qr)
        xp = spr.reconstruct(ap)
        outs.append((xz_sensors, y_qr, spr.Theta.copy(), ap, xp, spr.sensors_.copy()))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    assert rel_fro(outs[1][4][:, 0], x_test) < 0.1 or r < 4          # a sane reconstruction, not only equal ones
    return outs[1]


def run_deferred_reconstruct(eng):
    """ROM.defer_reconstruct (round 5, opt-in): reconstruct(wait=False) records its launch; the object's next fit() enqueues
    it in its host gap -- on the basis, centre and scale of the fit it was called AFTER, before the new projection overwrites
    them -- or wait() / any other method does.  Same values as the immediate form."""
    from openmeasure_amd.sparse_sensing import SPR

    def _np(t):
        return t.detach().cpu().numpy()
    rng = np.random.default_rng(8)
    n_points, F, m = 150, 2, 12
    X = rng.standard_normal((n_points * F, 5)) @ rng.standard_normal((5, m)) + 0.01 * rng.standard_normal((n_points * F, m))
    launches = []
    real = eng.reconstruct
    eng.reconstruct = lambda *a, **k: (launches.append(a[0].shape), real(*a, **k))[1]
    spr = SPR(X, F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=3)
    a3 = spr.Ar[:2].copy()
    want3 = spr.reconstruct(a3)                                  # immediate, host contract
    assert len(launches) == 1
    spr.defer_reconstruct = True
    pf = spr.reconstruct(a3, to_host=False, wait=False)
    assert pf.pending and not pf.launched and pf.shape == (2, n_points * F) and len(launches) == 1     # nothing enqueued yet
    spr.fit(select_modes='number', n_modes=5)                    # another basis (5 columns): the deferred launch precedes it
    assert pf.launched and len(launches) == 2 and launches[1] == (n_points * F, 3)
    np.testing.assert_array_equal(_np(pf.wait()).T, want3)
    assert spr.r == 5 and spr.Ur.shape == (n_points * F, 5)
    # wait() launches what no fit() has launched; to_host / wait=True never defer; other methods flush first
    pf2 = spr.reconstruct(spr.Ar[:1], to_host=False, wait=False)
    want5 = spr.reconstruct(spr.Ar[:1])                          # flushes pf2 first (order kept), then runs itself
    assert pf2.launched and len(launches) == 4
    np.testing.assert_array_equal(_np(pf2.wait()).T, want5)
    pf3 = spr.reconstruct(spr.Ar[:1], to_host=False, wait=False)
    spr.optimal_placement()
    assert pf3.launched
    pf4 = spr.reconstruct(spr.Ar[:1], to_host=False, wait=False)
    assert not pf4.launched
    np.testing.assert_array_equal(_np(pf4.wait()).T, want5)
    # a DEVICE tensor of coefficients is captured by value: what the caller writes into it after the call does not reach the launch
    a_dev = eng.to_device(spr.Ar[:1].copy())
    pf5 = spr.reconstruct(a_dev, to_host=False, wait=False)
    a_dev.mul_(3.0)
    assert not pf5.launched
    np.testing.assert_array_equal(_np(pf5.wait()).T, want5)
    assert isinstance(spr.reconstruct(spr.Ar[:1], to_host=False, wait=True), type(pf4.wait()))
