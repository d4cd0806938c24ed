"""Pins oracle/spr_oracle.py to the fixtures produced by the unmodified reference
(oracle/make_golden.py).  The oracle calls the same LAPACK routines as the reference,
so the comparison is bit-exact (assert_array_equal), not a tolerance."""
import numpy as np
import pytest

from oracle import spr_oracle as orc


def _fit(g):
    return orc.fit(g['X'], g['n_features'], g['select_modes'], g['n_modes'], scale_type=g['scale_type'],
                   axis_cnt=g['axis_cnt'])


def test_scale_data_bit_exact(golden):
    X_cnt, X_scl, X0 = orc.scale_data(golden['X'], golden['n_features'], golden['scale_type'], golden['axis_cnt'])
    np.testing.assert_array_equal(X_cnt, golden['X_cnt'])
    np.testing.assert_array_equal(X_scl, golden['X_scl'])
    if 'X0' in golden:
        np.testing.assert_array_equal(X0, golden['X0'])


def test_fit_bit_exact(golden):
    st = _fit(golden)
    assert st['r'] == golden['r']
    for k in ('Ur', 'Ar', 'Vr', 'Sigma_r', 'exp_variance'):
        np.testing.assert_array_equal(st[k], golden[k], err_msg=k)
    np.testing.assert_array_equal(st['S'], golden['S_full'])


def test_pivots_and_theta(golden):
    st = _fit(golden)
    mask = golden.get('mask')
    piv, Ur_m = orc.qr_pivots(st['Ur'], mask)
    np.testing.assert_array_equal(piv, golden['piv'])
    if mask is not None:
        np.testing.assert_array_equal(Ur_m, golden['Ur_after_placement'])
    n = golden['X'].shape[0]
    C = orc.one_hot_C(piv, n)
    assert tuple(golden['C_shape']) == C.shape
    Theta = orc.train_theta(C, Ur_m, n)
    np.testing.assert_array_equal(Theta, golden['Theta'])
    assert orc.theta_condition(Theta) == pytest.approx(float(golden['k']), rel=1e-13)


def test_predict_and_reconstruct(golden):
    g = golden
    st = _fit(g)
    n = g['X'].shape[0]
    n_points = n // g['n_features']
    piv, Ur_m = orc.qr_pivots(st['Ur'], g.get('mask'))
    C = orc.one_hot_C(piv, n)
    Theta = orc.train_theta(C, Ur_m, n)
    y0, cnt, scl = orc.scale_vector(g['ys'][1], C, st['X_cnt'], st['X_scl'], n_points)
    np.testing.assert_array_equal(y0, g['y0_1'])
    np.testing.assert_array_equal(cnt, g['cnt_vector'])
    np.testing.assert_array_equal(scl, g['scl_vector'])
    A1, S1 = orc.predict_ols(g['ys'][0], Theta, C, st['X_cnt'], st['X_scl'], n_points)
    A3, S3 = orc.predict_ols(list(g['ys']), Theta, C, st['X_cnt'], st['X_scl'], n_points)
    np.testing.assert_array_equal(A1, g['Ar_pred1'])
    np.testing.assert_array_equal(S1, g['Ar_sigma1'])
    np.testing.assert_array_equal(A3, g['Ar_pred3'])
    np.testing.assert_array_equal(S3, g['Ar_sigma3'])
    assert S3[1].any() and not S3[0].any()          # weighted vs unweighted branch both hit
    X1 = orc.reconstruct(A1[0], Ur_m, st['X_cnt'], st['X_scl'])
    X3 = orc.reconstruct(A3, Ur_m, st['X_cnt'], st['X_scl'])
    assert X1.shape == (n, 1) and X3.shape == (n, 3)
    np.testing.assert_array_equal(X1, g['X_rec1'])
    np.testing.assert_array_equal(X3, g['X_rec3'])
    S = g['sampling']
    np.testing.assert_array_equal(orc.reconstruct_sampled(A3, Ur_m, st['X_cnt'], st['X_scl'], S), g['X_rec3_sampled'])
    np.testing.assert_array_equal(orc.unscale_sampled(np.linspace(-1, 1, 7), S, st['X_cnt'], st['X_scl']),
                                  g['unscale_sampled'])


# ---- the reference's own unit tests (tests/test_rom.py, tests/test_spr.py), restated on
# ---- the oracle with a seeded matrix of the same shape (10 cells x 2 features x 5)
@pytest.fixture
def small():
    rng = np.random.default_rng(7)
    return rng.random((20, 5)), 2, 10


def test_ref_centering_and_scaling(small):            # test_rom.py:19-46
    X, F, npts = small
    X_cnt, X_scl, X0 = orc.scale_data_std(X, F)
    np.testing.assert_array_equal(X_cnt, np.mean(X, axis=1)[:, None])
    chk = np.zeros((20, 1))
    for f in range(F):
        chk[f * npts:(f + 1) * npts] = np.std(X[f * npts:(f + 1) * npts])
    np.testing.assert_array_equal(X_scl, chk)
    np.testing.assert_array_equal(X0, (X - np.mean(X, axis=1)[:, None]) / chk)
    X_cnt_none, _, _ = orc.scale_data_std(X, F, axis_cnt=None)   # test_rom.py:23-29
    for f in range(F):
        assert (X_cnt_none[f * npts:(f + 1) * npts] == np.mean(X[f * npts:(f + 1) * npts])).all()


def test_ref_decomposition_and_fit(small):             # test_rom.py:48-74
    X, F, _ = small
    _, _, X0 = orc.scale_data_std(X, F)
    U, S, Vt = np.linalg.svd(X0, full_matrices=False)
    Ur, Ar, _, _ = orc.decomposition(X0, n_modes=100)
    np.testing.assert_array_equal(U, Ur)
    np.testing.assert_array_equal(np.dot(np.diag(S), Vt).T, Ar)
    assert orc.decomposition(X0, 'number', 4)[1].shape[1] == 4
    st = orc.fit(X, F, n_modes=100)
    np.testing.assert_allclose(st['Vr'], Vt.T)
    np.testing.assert_allclose(st['Sigma_r'], S)


def test_ref_roundtrips(small):                        # test_rom.py:76-85, test_spr.py:21-60
    X, F, npts = small
    st = orc.fit(X, F, n_modes=100)
    np.testing.assert_allclose(orc.unscale(st['X0'][:, 0], st['X_cnt'], st['X_scl']), X[:, 0])
    np.testing.assert_allclose(orc.reconstruct(st['Ar'][0, :], st['Ur'], st['X_cnt'], st['X_scl']), X[:, [0]])
    piv, _ = orc.qr_pivots(st['Ur'])
    assert orc.one_hot_C(piv, 20).shape == (5, 20)
    C = np.eye(20)
    Theta = orc.train_theta(C, st['Ur'], 20)
    y = np.zeros((20, 3))
    y[:, 0] = X[:, 0]
    y[npts:, 2] = 1
    y0, _, _ = orc.scale_vector(y, C, st['X_cnt'], st['X_scl'], npts)
    chk = np.zeros((20, 2))
    chk[:, 0] = (y[:, 0] - st['X_cnt'][:, 0]) / st['X_scl'][:, 0]
    np.testing.assert_allclose(y0, chk)
    a, _ = orc.predict_ols(y, Theta, C, st['X_cnt'], st['X_scl'], npts)
    np.testing.assert_allclose(orc.reconstruct(a, st['Ur'], st['X_cnt'], st['X_scl']), X[:, [0]])


def test_error_paths():
    X = np.zeros((6, 3))
    with pytest.raises(TypeError):
        orc.check_inputs([[1.0]], 1)
    with pytest.raises(TypeError):
        orc.check_inputs(X, 2.0)
    with pytest.raises(Exception):
        orc.check_inputs(X, 4)
    ev = np.array([50.0, 90.0, 100.0])
    with pytest.raises(ValueError):
        orc.select_rank(ev, 3, 'variance', 101)
    with pytest.raises(TypeError):
        orc.select_rank(ev, 3, 'number', 2.0)
    with pytest.raises(ValueError):
        orc.select_rank(ev, 3, 'number', 4)
    with pytest.raises(ValueError):
        orc.select_rank(ev, 3, 'bogus', 1)
    assert orc.select_rank(ev, 3, 'variance', 99) == 3
    assert orc.select_rank(ev, 3, 'variance', 90) == 2
    assert orc.select_rank(ev, 3, 'variance', 100) == 3


def test_gem_pivots(golden_gem):                       # :586-698, noise-free limit == the reference under 3 noise seeds
    g = golden_gem
    st = orc.fit(g['X'], g['n_features'], 'number', g['n_modes'])
    np.testing.assert_array_equal(st['Ur'], g['Ur'])
    piv, lead = orc.gem_pivots(st['Ur'], g['n_sensors'], g['xyz'], g['n_features'], g.get('mask'), g['d_min'])
    np.testing.assert_array_equal(piv, g['gem_piv'])
    assert lead.min() > 1e-3                            # far above the 1e-5 regularisation noise of the reference
    rng = np.random.default_rng(3)
    piv_n, _ = orc.gem_pivots(st['Ur'], g['n_sensors'], g['xyz'], g['n_features'], g.get('mask'), g['d_min'],
                              noise=lambda k: 1e-5 * rng.standard_normal(k))
    np.testing.assert_array_equal(piv_n, g['gem_piv'])


def _oracle_predict_block(g, Ur, rows):
    n = g['X'].shape[0]
    n_points = n // g['n_features']
    C = np.zeros((len(rows), n)); C[np.arange(len(rows)), rows] = 1.0
    Theta = orc.train_theta(C, Ur, n)
    np.testing.assert_array_equal(Theta, g['Theta'])
    A3, S3 = orc.predict_ols(list(g['ys']), Theta, C, g['X_cnt'], g['X_scl'], n_points)
    np.testing.assert_array_equal(A3, g['Ar_pred3'])
    np.testing.assert_array_equal(S3, g['Ar_sigma3'])
    np.testing.assert_array_equal(orc.reconstruct(A3, Ur, g['X_cnt'], g['X_scl']), g['X_rec3'])


def test_gem_predict_minimum_norm(golden_gem):         # fewer sensors than modes: pinv's minimum-norm solution (:873-878)
    g = golden_gem
    assert g['n_sensors'] < g['Ur'].shape[1]
    _oracle_predict_block(g, g['Ur'], g['gem_piv'])


def test_pinv_cases(golden_pinv):                      # underdetermined / rank-deficient / ill-conditioned W Theta
    g = golden_pinv
    X_cnt, X_scl, _ = orc.scale_data(g['X'], g['n_features'])
    np.testing.assert_array_equal(X_cnt, g['X_cnt'])
    np.testing.assert_array_equal(X_scl, g['X_scl'])
    _oracle_predict_block(g, g['Ur'], g['C_rows'])
    sv = g['sv_WTheta']
    r = g['Ur'].shape[1]
    if g['kind'] == 'under':
        assert len(g['C_rows']) == r - 2
    if g['kind'] in ('dup', 'zerocol'):                 # numerically rank r-1: the smallest singular value is below the rcond cut
        assert (sv[:, -1] <= 1e-15 * sv[:, 0]).all() and (sv[:, -2] > 1e-6 * sv[:, 0]).all()
    if g['kind'] == 'illcond':
        assert (sv[:, 0] / sv[:, -1] > 3e6).all() and (sv[:, 0] / sv[:, -1] < 1e10).all()


def test_scale_limits():                               # :173-210, both the ordinary and the clamped branch
    import os
    from tests.conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'lim_small.npz'))
    F = int(g['n_features'])
    X_cnt, X_scl, _ = orc.scale_data(g['X'], F)
    l0 = orc.scale_limits([g['lo'], g['hi']], X_cnt, X_scl, F)
    l1 = orc.scale_limits([g['lo_far'], g['hi_far']], X_cnt, X_scl, F)
    np.testing.assert_array_equal(l0[0], g['lim0_lo']); np.testing.assert_array_equal(l0[1], g['lim0_hi'])
    np.testing.assert_array_equal(l1[0], g['lim1_lo']); np.testing.assert_array_equal(l1[1], g['lim1_hi'])
