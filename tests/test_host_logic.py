"""CPU tests of the host half of openmeasure_amd (class surface, validation, Gram-route
algebra, merges) with the NumPy test-double engine.  The HIP kernels are NOT exercised
here -- that is tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest
import scipy.sparse as sp

from openmeasure_amd.sparse_sensing import ROM, SPR
from tests.numpy_engine import NumpyEngine
from tests.parity import (run_conditioning_guard, run_documented_idioms, run_f32_storage, run_gem_beyond_rank, run_fixture, run_gem_fixture, run_gpr_style,
                          run_pinv_fixture)


def test_fixture_through_host_logic(golden):
    run_fixture(golden, NumpyEngine())


@pytest.mark.parametrize('foreign', [False, True])
def test_gpr_style_subclass(golden, foreign):           # gpr.py:379-402: a ROM subclass assigns X0 / Ur / Ar itself
    rom = run_gpr_style(golden, NumpyEngine(), foreign_basis=foreign)
    assert isinstance(rom.Ur, np.ndarray)


def test_gem_fixture_through_host_logic(golden_gem):   # :586-698
    run_gem_fixture(golden_gem, NumpyEngine())


@pytest.mark.parametrize('n_points,F,r,n_sensors,d_min,masked', [(150, 2, 5, 9, 0.0, False), (200, 3, 6, 8, 0.08, True),
                                                              (120, 2, 4, 20, 0.0, False)])
def test_gem_beyond_rank_through_host_logic(n_points, F, r, n_sensors, d_min, masked):
    run_gem_beyond_rank(NumpyEngine(), n_points, F, r, n_sensors, d_min, masked, 5 + r)


def test_pinv_fixture_through_host_logic(golden_pinv):  # :873-878 minimum-norm / rank-deficient / ill-conditioned
    run_pinv_fixture(golden_pinv, NumpyEngine())


def _synth(n_points, F, m, k, rho, eps, seed):
    rng = np.random.default_rng(seed)
    n = n_points * F
    X = rng.standard_normal((n, k)) @ ((rho ** np.arange(k))[:, None] * rng.standard_normal((k, m)))
    X += eps * rng.standard_normal((n, m))
    for f in range(F):
        X[f * n_points:(f + 1) * n_points] = (f + 1) * X[f * n_points:(f + 1) * n_points] + 10.0 * f
    return np.ascontiguousarray(X)


@pytest.mark.parametrize('decades', [3, 6, 9, 12])
def test_conditioning_guard_through_host_logic(decades):   # SURVEY 7 hard part 1: exact sensors or explicit refusal
    run_conditioning_guard(NumpyEngine(), decades, _synth)
    if decades == 6:
        run_conditioning_guard(NumpyEngine(), decades, _synth, f32=True)


@pytest.mark.parametrize('decades', [5, 6, 7])
def test_conditioning_guard_wide_matrix_takes_the_top_r_svd(decades, monkeypatch):
    """m >= 96 with r <= m / 2: the refinement pass ends with spr_host_svd_top (all singular values, the r retained right vectors)
    instead of the full dgesdd -- same sensors, spectrum and basis as the oracle; and the same with the route switched off."""
    from openmeasure_amd import _eigen as ss
    calls = []
    real = ss._svd_top_native
    monkeypatch.setattr(ss, '_svd_top_native', lambda M, r: calls.append(M.shape) or real(M, r))
    spr = run_conditioning_guard(NumpyEngine(), decades, _synth, shape=(500, 2, 160, 24))
    assert spr.gram_refine_passes_ >= 1 and calls and calls[0] == (160, 160)
    monkeypatch.setattr(ss, '_svd_top_native', lambda M, r: None)
    run_conditioning_guard(NumpyEngine(), decades, _synth, shape=(500, 2, 160, 24))


@pytest.mark.parametrize('m,r,decades', [(128, 32, 5.0), (256, 64, 6.1), (300, 40, 8.0), (96, 48, 4.0), (200, 1, 3.0)])
def test_svd_top_native_against_lapack(m, r, decades):
    """spr_host_svd_top on the factor a refinement pass hands it (R diag(d) V^T, R the Cholesky factor of a matrix near I): all
    singular values to the relative accuracy of LAPACK's own, the r leading right singular vectors up to sign, orthonormal."""
    from openmeasure_amd._eigen import _svd_top_native
    rng = np.random.default_rng(m + r)
    V, _ = np.linalg.qr(rng.standard_normal((m, m)))
    d = np.concatenate([np.logspace(0, -decades, r), 10 ** (-decades - 0.5) * (1 + 0.1 * rng.random(m - r))])
    E = 1e-3 * rng.standard_normal((m, m))
    R = np.linalg.cholesky(np.eye(m) + 0.5 * (E + E.T)).T
    M = (R * d[None, :]) @ V.T
    out = _svd_top_native(M, r)
    assert out is not None
    S, Vr = out
    _, S_ref, Vt = np.linalg.svd(M)
    np.testing.assert_allclose(S[:r], S_ref[:r], rtol=1e-11)
    np.testing.assert_allclose(S, S_ref, rtol=0, atol=1e-13 * S_ref[0])           # the noise floor: to LAPACK's ABSOLUTE accuracy
    assert np.abs(Vr.T @ Vr - np.eye(r)).max() <= 1e-12
    gap = np.min(np.abs(np.diff(S_ref[:r + 1]))) / S_ref[0] if r < m else 1.0
    assert np.max(1.0 - np.abs(np.sum(Vr * Vt[:r].T, axis=0))) <= 1e-13 / max(gap, 1e-3) ** 2 + 1e-12


def test_svd_top_native_declines():
    """None (the caller takes np.linalg.svd) for small matrices, r above m / 2, non-finite input -- and for a cluster of equal
    singular values among the r leading ones, where inverse iteration without re-orthogonalisation cannot separate the vectors."""
    from openmeasure_amd._eigen import _svd_top_native
    rng = np.random.default_rng(3)
    assert _svd_top_native(rng.standard_normal((40, 40)), 8) is None
    assert _svd_top_native(rng.standard_normal((128, 128)), 65) is None
    bad = rng.standard_normal((128, 128))
    bad[3, 4] = np.nan
    assert _svd_top_native(bad, 8) is None
    Q1, _ = np.linalg.qr(rng.standard_normal((128, 128)))
    Q2, _ = np.linalg.qr(rng.standard_normal((128, 128)))
    s = np.logspace(0, -3, 128)
    s[4:8] = s[4]                                                                  # a four-fold singular value
    assert _svd_top_native((Q1 * s) @ Q2.T, 16) is None


@pytest.mark.parametrize('n_points,F,m,r', [(400, 3, 12, 4), (300, 2, 41, 14)])
def test_f32_storage_through_host_logic(n_points, F, m, r):      # float32 X: stored as given, arithmetic in f64
    run_f32_storage(NumpyEngine(), n_points, F, m, r, 21, _synth)


@pytest.fixture
def small():
    rng = np.random.default_rng(11)
    return rng.random((20, 5)), 2, rng.random((10, 3))


def test_constructor_errors(small):                    # sparse_sensing.py:69-81
    X, F, xyz = small
    with pytest.raises(TypeError):
        ROM(X.tolist(), F, xyz)
    with pytest.raises(TypeError):
        ROM(X, 2.0, xyz)
    with pytest.raises(Exception):
        ROM(X, 3, xyz)
    rom = ROM(X, F, xyz)
    assert rom.n_points == 10 and rom.X is X           # keeps a reference, no copy (:74)


def test_mode_selection_errors(small):                 # :314-333
    X, F, xyz = small
    rom = ROM(X, F, xyz, engine=NumpyEngine())
    with pytest.raises(ValueError):
        rom.fit(select_modes='variance', n_modes=101)
    with pytest.raises(TypeError):
        rom.fit(select_modes='number', n_modes=2.0)
    with pytest.raises(ValueError):
        rom.fit(select_modes='number', n_modes=6)
    with pytest.raises(ValueError):
        rom.fit(select_modes='bogus')
    rom.fit(select_modes='number', n_modes=4)
    assert rom.r == 4
    rom.fit(select_modes='variance', n_modes=100)
    assert rom.r == 5


def test_unsupported_options_raise_not_fallback(small):
    X, F, xyz = small
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    # 'vast_2/3/4' (:147-157) put scipy's per-COLUMN kurtosis (m values) into a slice of n_points rows: the reference itself
    # fails with NumPy's broadcast ValueError unless m == n_points (or m == 1) -- same type and text here (checked against
    # the imported reference: 'could not broadcast input array from shape (5,) into shape (10,)')
    for st in ('vast_2', 'vast_3', 'vast_4'):
        with pytest.raises(ValueError, match=r'could not broadcast input array from shape \(5,\) into shape \(10,\)'):
            spr.fit(scale_type=st)
    sq = SPR(np.random.default_rng(2).random((10, 5)) + 1, 2, None, engine=NumpyEngine())      # n_points == m: defined there,
    with pytest.raises(NotImplementedError):                                                   # no device implementation here
        sq.fit(scale_type='vast_2')
    with pytest.raises(NotImplementedError):
        spr.fit(scale_type='bogus')                    # :164
    # axis_cnt goes to np.average(x, axis=axis_cnt) whose result is assigned to n_points rows (:112) -- checked against the
    # imported reference: 0 / -2 -> ValueError (broadcast), 2 / 7 / -3 -> AxisError, -1 = 1 (round 6)
    for ax in (0, -2):
        with pytest.raises(ValueError, match=r'could not broadcast input array from shape \(5,\) into shape \(10,\)'):
            spr.fit(axis_cnt=ax)
    for ax in (2, 7, -3):
        with pytest.raises(np.exceptions.AxisError, match=f'axis {ax} is out of bounds for array of dimension 2'):
            spr.fit(axis_cnt=ax)
    spr.fit(axis_cnt=-1, n_modes=100)
    xc = spr.X_cnt.copy()
    spr.fit(axis_cnt=1, n_modes=100)
    np.testing.assert_array_equal(xc, spr.X_cnt)
    with pytest.raises(NotImplementedError):
        sq.fit(axis_cnt=0)                             # m == n_points: defined in the reference (column means), not here
    spr.fit(n_modes=100)
    Cg = spr.optimal_placement(calc_type='gem', n_sensors=spr.r + 2)  # > r-1: deterministic ridge stand-in for the noise
    assert Cg.shape == (spr.r + 2, 20) and len(set(spr.sensors_.tolist())) == spr.r + 2
    C0 = spr.optimal_placement(calc_type='gem', n_sensors=0)                 # the reference loops range(n_sensors): an empty placement
    assert C0.shape == (0, 20) and spr.sensors_.shape == (0,)
    with pytest.raises(TypeError, match="'float' object cannot be interpreted as an integer"):
        spr.optimal_placement(calc_type='gem', n_sensors=2.0)
    with pytest.raises(NotImplementedError):
        spr.optimal_placement(calc_type='bogus')       # :752-754
    with pytest.raises(NotImplementedError):
        spr.train(np.eye(20), method='COLS')
    with pytest.raises(ValueError, match=r'shapes \(19,19\) and \(20,1\) not aligned: 19 \(dim 1\) != 20 \(dim 0\)'):   # NumPy's text (:366)
        spr.reconstruct(np.zeros(5), sampling=np.eye(19))
    with pytest.raises(ValueError, match=r'matmul: Input operand 1 has a mismatch in its core dimension 0.*size 20 is different from 19'):
        spr.unscale_data(np.zeros(19), sampling=np.eye(19))                                                               # (:233)


@pytest.mark.parametrize('n_points,F,m,seed', [(7, 2, 3, 0), (10, 3, 4, 1), (33, 1, 5, 2), (64, 4, 1, 3)])
def test_median_scaling_by_radix_selection(n_points, F, m, seed):   # :140-141, odd and even block sizes, ties, signs
    rng = np.random.default_rng(seed)
    X = np.round(rng.standard_normal((n_points * F, m)) * 3, 1) + rng.integers(-2, 3, size=(n_points * F, 1)) + 0.05
    X[0, 0] = -0.0
    rom = ROM(X, F, None, engine=NumpyEngine())
    rom.scale_data('median')
    want = np.array([np.median(X[f * n_points:(f + 1) * n_points]) for f in range(F)])
    np.testing.assert_array_equal(rom._scl_f, want)
    np.testing.assert_array_equal(rom.X_scl[:, 0], np.repeat(want, n_points))


def test_scale_limits_matches_reference_fixture():      # :173-210
    import os
    from tests.conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, 'lim_small.npz'))
    rom = ROM(g['X'].copy(), int(g['n_features']), None, engine=NumpyEngine())
    rom.fit(select_modes='number', n_modes=3)
    for (lo, hi, want_lo, want_hi) in ((g['lo'], g['hi'], g['lim0_lo'], g['lim0_hi']),
                                       (g['lo_far'], g['hi_far'], g['lim1_lo'], g['lim1_hi'])):
        got = rom.scale_limits([lo, hi])
        np.testing.assert_allclose(got[0], want_lo, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(got[1], want_hi, rtol=1e-12, atol=1e-12)


def test_empty_batches(small):                          # :863-864, :362-373 with n_p = 0
    X, F, xyz = small
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    spr.fit(select_modes='number', n_modes=3)
    spr.train(spr.optimal_placement())
    Ar, Ar_sigma = spr.predict([])
    assert Ar.shape == (0, 3) and Ar_sigma.shape == (0, 3)
    assert spr.reconstruct(np.zeros((0, 3))).shape == (20, 0)


def test_train_predict_errors(small):                  # :791-793, :801-803, :848-854
    X, F, xyz = small
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    spr.fit(select_modes='number', n_modes=3)
    with pytest.raises(ValueError):
        spr.train(np.eye(19))
    with pytest.raises(ValueError):
        spr.train(np.zeros((4, 2)), is_Theta=True)
    with pytest.raises(AttributeError):
        spr.predict(np.zeros((3, 3)))                  # predict before train: no Theta
    C = spr.optimal_placement()
    spr.train(C)
    with pytest.raises(ValueError):
        spr.predict(np.zeros((4, 3)))
    with pytest.raises(ValueError):
        spr.predict(np.zeros((3, 2)))
    spr.method = 'bogus'
    with pytest.raises(NotImplementedError):           # :894-896
        spr.predict(np.zeros((3, 3)))
    spr2 = SPR(X, F, xyz, engine=NumpyEngine())
    spr2.fit(select_modes='number', n_modes=3)
    spr2.train(np.zeros((3, 3)), is_Theta=True)
    with pytest.raises(AttributeError):                # no C -> scale_vector cannot run (:573)
        spr2.predict(np.zeros((3, 3)))


def test_reference_unit_tests_on_class(small):         # tests/test_rom.py, tests/test_spr.py restated
    X, F, xyz = small
    n_points = 10
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    X0 = spr.scale_data()
    np.testing.assert_allclose(spr.X_cnt, np.mean(X, axis=1)[:, None], rtol=1e-14)
    scl = np.zeros((20, 1))
    for f in range(F):
        scl[f * n_points:(f + 1) * n_points] = np.std(X[f * n_points:(f + 1) * n_points])
    np.testing.assert_allclose(spr.X_scl, scl, rtol=1e-13)
    np.testing.assert_allclose(X0, (X - np.mean(X, axis=1)[:, None]) / scl, rtol=1e-12, atol=1e-14)
    spr.fit(n_modes=100)
    assert spr.r == 5
    _, S, Vt = np.linalg.svd(X0, full_matrices=False)
    np.testing.assert_allclose(spr.Sigma_r[:4], S[:4], rtol=1e-9)
    np.testing.assert_allclose(np.abs(np.sum(spr.Vr[:, :4] * Vt.T[:, :4], axis=0)), 1.0, rtol=1e-9)
    np.testing.assert_allclose(spr.unscale_data(X0[:, 0]), X[:, 0])
    np.testing.assert_allclose(spr.reconstruct(spr.Ar[0, :]), X[:, [0]])            # test_rom.py:82-85
    Cq = spr.optimal_placement()
    assert Cq.shape == (5, 20)                                                         # test_spr.py:21-25
    C = np.eye(20)
    spr.train(C)
    y = np.zeros((20, 3))
    y[:, 0] = C @ X[:, 0]
    y[n_points:, 2] = 1
    y0 = spr.scale_vector(y)
    chk = np.zeros((20, 2))
    chk[:, 0] = (y[:, 0] - np.mean(X, axis=1)) / scl[:, 0]
    np.testing.assert_allclose(y0, chk, atol=1e-13)                                    # test_spr.py:27-46
    a, _ = spr.predict(y)
    np.testing.assert_allclose(spr.reconstruct(a), X[:, [0]])                          # test_spr.py:48-60


@pytest.mark.parametrize('stage', ['constructed', 'fitted', 'trained'])
def test_pickle_and_deepcopy_round_trip(small, stage):
    """The reference's objects are plain attributes and pickle as they are; here the state lives behind an engine: the pickle
    carries host copies of it (no engine, no device handles) and the loaded object uploads them on first use -- same
    attributes, same predictions, and it can be fitted again."""
    import copy
    import pickle
    from openmeasure_amd.rom import _DeviceState
    X, F, xyz = small
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    if stage != 'constructed':
        spr.fit(select_modes='number', n_modes=4)
    if stage == 'trained':
        C = spr.optimal_placement()
        spr.train(C)
    for clone in (pickle.loads(pickle.dumps(spr)), copy.deepcopy(spr)):
        assert clone._eng is None and isinstance(clone._d, _DeviceState) and len(clone._d) == 0     # nothing uploaded yet
        clone._eng = NumpyEngine()
        np.testing.assert_array_equal(clone.X, X)
        if stage == 'constructed':
            with pytest.raises(AttributeError):
                clone.Ur
        else:
            for name in ('Ur', 'Ar', 'Vr', 'Sigma_r', 'X_cnt', 'X_scl'):
                np.testing.assert_array_equal(getattr(clone, name), getattr(spr, name), err_msg=name)
            assert clone.r == spr.r == 4
            np.testing.assert_array_equal(clone.reconstruct(clone.Ar[0, :]), spr.reconstruct(spr.Ar[0, :]))
        if stage == 'trained':
            np.testing.assert_array_equal(clone.sensors_, spr.sensors_)
            y = np.zeros((4, 3))
            y[:, 0] = X[spr.sensors_, 1]
            y[:, 2] = spr.sensors_ // 10
            a0, s0 = spr.predict(y)
            a1, s1 = clone.predict(y)
            np.testing.assert_array_equal(a1, a0)
            np.testing.assert_array_equal(clone.reconstruct(a1), spr.reconstruct(a0))
        clone.fit(select_modes='number', n_modes=3)                    # a loaded object is a full object
        assert clone.r == 3 and clone.Ur.shape == (20, 3)


def test_public_gem_method():
    """SPR.gem(Ur, n_sensors, mask, d_min, verbose) -- the method optimal_placement('gem') calls (reference :586-698, :747) --
    on the fitted basis, on a foreign basis (fitted state untouched) and on an object that was never fitted; against the
    oracle's literal covariance formulas."""
    from oracle import spr_oracle as orc
    rng = np.random.default_rng(1)
    n_points, F, r = 80, 2, 6
    X, xyz = rng.standard_normal((n_points * F, 9)), rng.random((n_points, 2))
    U = rng.standard_normal((n_points * F, r))
    mask = rng.random(n_points * F) < 0.6
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    want, _ = orc.gem_pivots(U, 4, xyz, F, None, 0.0)
    np.testing.assert_array_equal(spr.gem(U, 4, None, 0.0, False), want)            # never fitted: only Ur and xyz matter
    assert not hasattr(spr, 'r') and not hasattr(spr, 'sensors_')
    spr.fit(select_modes='number', n_modes=r)
    C = spr.optimal_placement(calc_type='gem', n_sensors=4)
    np.testing.assert_array_equal(spr.gem(spr.Ur, 4, None, 0.0, False), np.argmax(np.asarray(C), axis=1))
    Ur0, sensors0 = spr.Ur.copy(), spr.sensors_.copy()
    want, _ = orc.gem_pivots(U, 5, xyz, F, mask, 0.05)
    np.testing.assert_array_equal(spr.gem(U, 5, mask, 0.05, True), want)
    assert spr.r == r
    np.testing.assert_array_equal(spr.Ur, Ur0)
    np.testing.assert_array_equal(spr.sensors_, sensors0)
    with pytest.raises(ValueError):
        spr.gem(U[:-1], 4, None, 0.0, False)


def test_gem_depends_on_the_signs_of_the_basis_and_prints_the_reference_table(capsys):
    """The GEM rule takes the variance of a row of Ur over its r entries (reference :622, :638): flipping the sign of ONE
    column of the basis -- something LAPACK is free to do -- moves the sensors.  The dependence is the algorithm's: the
    class and the oracle's literal formulas move to the SAME new sensors.  (Hence: fit() -> optimal_placement('gem') gives
    the GEM sensors of this implementation's basis; the reference's only after fit(basis=(Ur_ref, Ar_ref)).)
    verbose=True prints the reference's table (:633-635, :652, :694)."""
    from oracle import spr_oracle as orc
    rng = np.random.default_rng(5)
    n_points, F, r = 400, 2, 8
    n = n_points * F
    U = rng.standard_normal((n, r)) * (0.5 + rng.random((n, 1))) / np.sqrt(n)
    xyz = rng.random((n_points, 3))
    U2 = U.copy()
    U2[:, 2] *= -1
    spr = SPR(np.zeros((n, r + 1)), F, xyz, engine=NumpyEngine())
    got, got2 = spr.gem(U, 6, None, 0.0, False), spr.gem(U2, 6, None, 0.0, True)
    want, lead = orc.gem_pivots(U, 6, xyz, F, None, 0.0)
    want2, lead2 = orc.gem_pivots(U2, 6, xyz, F, None, 0.0)
    assert min(lead.min(), lead2.min()) > 1e-6
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got2, want2)
    assert not np.array_equal(want, want2)                     # the sign of one column changed the sensors
    out = capsys.readouterr().out.splitlines()
    assert '# sensors' in out[1] and 'sigma^2 y|a' in out[1] and 'Htot' in out[1]
    rows = [ln.split() for ln in out if ln.strip() and ln.split()[0].isdigit()]
    assert [int(t[0]) for t in rows] == [1, 2, 3, 4, 5, 6] and rows[0][2:] == ['-', '-']
    # the numbers are the reference's quantities: row variance and conditional variance of every pick in its scaled units
    coef = 2 / np.sqrt(np.var(U2, ddof=1, axis=1).max())
    A = U2[want2] * coef
    np.testing.assert_allclose([float(t[1]) for t in rows], np.var(A, ddof=1, axis=1), rtol=6e-3)
    S = np.cov(A[:3], ddof=1)
    full = np.cov(A[:3], A[3], ddof=1)
    cond = full[-1, -1] - full[-1, :-1] @ np.linalg.inv(S) @ full[:-1, -1]
    assert abs(float(rows[3][2]) - cond) <= 6e-3 * cond
    assert all(float(a[3]) != float(b[3]) for a, b in zip(rows[1:], rows[2:]))    # entropy accumulates


def test_deferred_reconstruct_runs_in_the_next_fits_gap():
    from tests.parity import run_deferred_reconstruct
    run_deferred_reconstruct(NumpyEngine())


@pytest.mark.parametrize('m,r', [(32, 8), (41, 14), (64, 32), (95, 40)])
def test_native_top_r_eigen_route(m, r):
    """spr_host_eig_top (round 5): dsytrd + dsterf + batched inverse iterations + dormtr in ONE host call of the library, LAPACK
    reached through SciPy's exported function pointers -- against dsyevd; a cluster of equal eigenvalues makes it decline (the
    Python route with dstein / dsyevd takes over); fit() takes it for 32 <= m < 96 when the number of modes is given."""
    import openmeasure_amd._eigen as ss
    rng = np.random.default_rng(m)
    A = rng.standard_normal((6 * m, m)) * (0.9 ** np.arange(m))
    A -= A.mean(axis=1, keepdims=True)
    G = A.T @ A
    got = ss._eig_top_native(G, r)
    if got is None:
        pytest.skip('library or LAPACK pointers not available')
    lam, V = got
    w, vv = ss._eigh_small(G)
    w, vv = w[::-1], vv[:, ::-1][:, :r]
    np.testing.assert_allclose(lam, w, rtol=0, atol=1e-13 * w[0])
    assert np.abs(V.T @ V - np.eye(r)).max() <= 1e-12
    assert np.abs(G @ V - V * lam[:r]).max() <= 1e-13 * lam[0]
    sg = np.sign(np.sum(V * vv, axis=0))
    np.testing.assert_allclose(V * sg, vv, rtol=0, atol=1e-7)
    # r equal eigenvalues: inverse iteration without re-orthogonalisation cannot separate them -> declined, never wrong
    Q = np.linalg.qr(rng.standard_normal((m, m)))[0]
    lamc = np.concatenate([np.full(r, 5.0), np.linspace(1.0, 0.1, m - r)])
    assert ss._eig_top_native((Q * lamc) @ Q.T, r) is None
    # ... and fit() uses it: same basis as with the route switched off
    from tests.numpy_engine import NumpyEngine
    X = rng.standard_normal((300, 6)) @ rng.standard_normal((6, m)) + 0.01 * rng.standard_normal((300, m))
    a = SPR(X, 3, None, engine=NumpyEngine()); a.fit(select_modes='number', n_modes=4)
    old = ss._EIGH_TOP_NATIVE_MIN_M
    try:
        ss._EIGH_TOP_NATIVE_MIN_M = 10 ** 6
        b = SPR(X, 3, None, engine=NumpyEngine()); b.fit(select_modes='number', n_modes=4)
    finally:
        ss._EIGH_TOP_NATIVE_MIN_M = old
    np.testing.assert_allclose(a.Sigma_r, b.Sigma_r, rtol=1e-12)
    np.testing.assert_allclose(np.abs(a.Ur), np.abs(b.Ur), atol=1e-9)


def test_decomposition_public(small):
    X, F, xyz = small
    rom = ROM(X, F, xyz, engine=NumpyEngine())
    X0 = rom.scale_data()
    Ur, Ar, ev = rom.decomposition(X0, select_modes='number', n_modes=4)
    U, S, Vt = np.linalg.svd(X0, full_matrices=False)
    assert rom.r == 4 and Ur.shape == (20, 4) and Ar.shape == (5, 4)
    np.testing.assert_allclose(np.abs(np.sum(Ur * U[:, :4], axis=0)), 1.0, rtol=1e-9)
    np.testing.assert_allclose(Ur @ Ar.T, (U[:, :4] * S[:4]) @ Vt[:4], atol=1e-10)
    np.testing.assert_allclose(ev, (100 * np.cumsum(S ** 2) / np.sum(S ** 2))[:4])


def test_train_accepts_sparse_and_dense_general_C(small):
    X, F, xyz = small
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    spr.fit(select_modes='number', n_modes=3)
    rng = np.random.default_rng(5)
    C = rng.random((6, 20)) * (rng.random((6, 20)) < 0.3)
    spr.train(C)
    np.testing.assert_allclose(spr.Theta, C @ spr.Ur, atol=1e-13)
    T1 = spr.Theta.copy()
    spr.train(sp.csr_matrix(C))
    np.testing.assert_allclose(spr.Theta, T1, atol=1e-15)


def test_fit_with_given_basis(small):                  # :493-497
    X, F, xyz = small
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    spr.fit(select_modes='number', n_modes=3)
    Ur, Ar = spr.Ur.copy(), spr.Ar.copy()
    spr2 = SPR(X, F, xyz, engine=NumpyEngine())
    spr2.fit(basis=(Ur, Ar))
    assert spr2.r == 3
    np.testing.assert_allclose(spr2.Sigma_r, spr.Sigma_r)
    np.testing.assert_allclose(spr2.reconstruct(Ar[1]), spr.reconstruct(Ar[1]))


@pytest.mark.parametrize('name', ['g2_num4', 'g3_num8', 'f32_g2_num4'])
def test_documented_idioms_on_one_hot_rows(name, monkeypatch):        # README.md:160-184, INTEGRATION.md
    from tests.conftest import load_golden
    run_documented_idioms(load_golden(name), NumpyEngine(), monkeypatch)


def test_one_hot_rows_matrix_semantics():
    from openmeasure_amd.sparse_sensing import OneHotRows
    rng = np.random.default_rng(3)
    rows = np.array([7, 2, 11, 2])
    C, D = OneHotRows(rows, 12), np.zeros((4, 12))
    D[np.arange(4), rows] = 1.0
    x, Xm = rng.standard_normal(12), rng.standard_normal((12, 3))
    np.testing.assert_array_equal(np.asarray(C), D)
    np.testing.assert_array_equal(C.toarray(), D)
    np.testing.assert_array_equal(C.tocsr().toarray(), D)
    np.testing.assert_array_equal(C @ x, D @ x)
    np.testing.assert_array_equal(C.dot(Xm), D.dot(Xm))
    np.testing.assert_array_equal(np.eye(4) @ C, D)
    np.testing.assert_array_equal(np.argmax(C, axis=1), np.argmax(D, axis=1))
    assert np.argmax(C) == np.argmax(D) and np.argmax(C[2, :]) == 11 and np.argmax(C[2]) == 11
    assert C[1, 2] == 1.0 and C[1, 3] == 0.0 and C[-1, :].argmax() == 2
    assert C.shape == D.shape and len(C) == 4 and C.ndim == 2 and C[0, :].shape == (12,) and C.dtype == D.dtype
    np.testing.assert_array_equal(np.asarray(C[1:3]), D[1:3])
    np.testing.assert_array_equal(np.asarray(C[0, :]), D[0, :])
    np.testing.assert_array_equal(C[:, 2:9], D[:, 2:9])
    np.testing.assert_array_equal(C.sum(axis=1), D.sum(axis=1))
    np.testing.assert_array_equal(C.sum(axis=0), D.sum(axis=0))
    assert C.sum() == D.sum() == 4
    with pytest.raises(ValueError):
        C @ np.ones(5)
    with pytest.raises(IndexError):
        C[4, :]
    with pytest.raises(IndexError):
        OneHotRows([12], 12)
    big = OneHotRows([5, 10 ** 9], 2 * 10 ** 9)                     # config-5-sized: nothing dense may be built
    with pytest.raises(MemoryError):
        np.asarray(big)
    assert np.argmax(big[1, :]) == 10 ** 9 and big.tocsr().shape == (2, 2 * 10 ** 9)


def test_state_read_before_fit_raises_attribute_error(small):        # the reference's attributes do not exist yet
    X, F, xyz = small
    spr = SPR(X, F, xyz, engine=NumpyEngine())
    for attr in ('Ur', 'X_cnt', 'X_scl', 'X0', 'Ar', 'Sigma_r', 'Vr'):
        with pytest.raises(AttributeError):
            getattr(spr, attr)
    with pytest.raises(AttributeError):
        spr.optimal_placement()
    with pytest.raises(AttributeError):
        spr.reconstruct(np.ones(3))
    with pytest.raises(AttributeError):
        spr.train(np.zeros((2, X.shape[0])))
    with pytest.raises(AttributeError):
        spr.unscale_data(np.ones(X.shape[0]))


@pytest.mark.parametrize('m', [12, 40])
def test_constant_feature_and_nan_raise_linalgerror(m):            # :169 -> :272: nan X0 -> 'SVD did not converge'
    rng = np.random.default_rng(m)
    X = rng.standard_normal((60, m))
    Xc = X.copy(); Xc[30:] = 3.0          # feature 1 constant (a value whose mean is exact): X_scl = 0 exactly, X0 = 0/0
    Xn = X.copy(); Xn[7, 3] = np.nan
    Xi = X.copy(); Xi[41, 0] = np.inf
    for bad in (Xc, Xn, Xi):
        spr = SPR(bad, 2, None, engine=NumpyEngine())
        with pytest.raises(np.linalg.LinAlgError):
            with np.errstate(all='ignore'):
                spr.fit(select_modes='number', n_modes=3)


def test_partial_row_group_with_absent_features():
    """RowShard(partial=True): a block that holds rows of features 0 and 1 of F = 4 -- features 2 and 3 have no rows, take
    no part (scale 1) and must not turn the Gram matrix into 0/0 (the merge on the host; csrc/combine.hip on the GPU)."""
    from openmeasure_amd.sparse_sensing import RowShard
    n_points, F, m, r = 900, 4, 40, 6
    X = _synth(n_points, F, m, 12, 0.7, 1e-3, 31)
    row0, n_loc = 300, 1200
    blk = np.ascontiguousarray(X[row0:row0 + n_loc])
    for scale_type in ('std', 'range'):                                 # device-merge scalings and the host-merge ones
        spr = SPR(blk, F, None, shard=RowShard(row0, n_points * F, partial=True), engine=NumpyEngine())
        spr.fit(scale_type=scale_type, select_modes='number', n_modes=r)
        assert np.isfinite(spr.S_).all() and np.isfinite(spr.Ur).all()
        np.testing.assert_array_equal(spr._scl_f[2:], 1.0)
    a, b = blk[:600], blk[600:]
    Xc = np.vstack([(a - a.mean(1, keepdims=True)) / (a.max() - a.min()), (b - b.mean(1, keepdims=True)) / (b.max() - b.min())])
    np.testing.assert_allclose(spr.S_[:r], np.linalg.svd(Xc, compute_uv=False)[:r], rtol=1e-9)


def test_gem_ridge_phase_with_dependent_picks():
    """calc_type='gem' beyond r-1 sensors when the first r-1 picks are numerically dependent (rows of Ur that span only
    part of the centred space, with exact duplicates): the ridge phase must neither raise nor pick a row twice."""
    rng = np.random.default_rng(8)
    n_points, F, r = 60, 2, 6
    n = n_points * F
    B = rng.standard_normal((3, r))                                  # rows of Ur live in a 3-dimensional subspace
    Ur = rng.standard_normal((n, 3)) @ B
    Ur[10] = Ur[3]; Ur[77] = Ur[3]                                   # exact duplicates
    spr = SPR(rng.standard_normal((n, 4)), F, rng.random((n_points, 3)), engine=NumpyEngine())
    spr.fit(basis=(Ur.copy(), np.eye(4, r)))
    C = spr.optimal_placement(calc_type='gem', n_sensors=r + 3)
    assert C.shape == (r + 3, n)
    assert len(set(spr.sensors_.tolist())) == r + 3 and spr.sensors_.min() >= 0


# ---- the placement drivers on a candidate-set model of the device protocol (tests/numpy_engine.py: CandidateEngine) ----
def _pool_cases():
    rng = np.random.default_rng(0)
    out = []
    for n, r in ((600, 12), (900, 20), (300, 8), (2000, 24), (257, 16)):
        U = rng.standard_normal((n, r))
        out.append((f'heavy-{n}x{r}', U * np.exp(1.5 * rng.standard_normal((n, 1)))))     # few rows far above the rest
        out.append((f'uniform-{n}x{r}', np.linalg.qr(U)[0]))                             # pools never pay
    U = rng.standard_normal((400, 10)) * np.exp(rng.standard_normal((400, 1)))
    U[100:140] = U[60:100]                                                              # duplicated rows: ties
    U[300:330] = 0.0                                                                    # zero rows (a mask)
    out.append(('ties-and-zero-rows', U))
    return out


@pytest.mark.parametrize('name,U', _pool_cases(), ids=[c[0] for c in _pool_cases()])
@pytest.mark.parametrize('pools', [False, True])
def test_pivot_drivers_on_the_candidate_model(name, U, pools):
    """pivot_loop / _pivot_loop_pooled against dgeqp3's order, on a NumPy model of the candidate-set protocol whose blocks
    are small enough (16 rows, 2 candidates each, 8 directions per epoch sweep) that batches fail their certification,
    pools empty out and full sweeps have to step in."""
    from openmeasure_amd.sparse_sensing import pivot_loop
    from oracle import spr_oracle as orc
    from tests.numpy_engine import CandidateEngine
    r = U.shape[1]
    ref, _ = orc.qr_pivots(U)
    eng = CandidateEngine()
    st = eng.qr_begin(eng.to_device(U), 0, r)
    stats = {}
    sweeps = pivot_loop(eng, st, r, pools=pools, stats=stats)
    piv = st['piv'].numpy()
    if name.startswith('ties'):      # equal norms: LAPACK's and our tie rule agree on the first of equals only while norms are exact
        assert len(set(piv.tolist())) == r
        k = int(np.argmax(piv != ref)) if (piv != ref).any() else r
        assert k >= 1
    else:
        np.testing.assert_array_equal(piv, ref)
    kinds = [e[0] for e in eng.log]
    if pools:
        assert 'refresh' not in kinds and sweeps == 1 + kinds.count('full') and stats['pool_sweeps'] == kinds.count('pool')
        if name.startswith('heavy'):
            eng2 = CandidateEngine()
            st2 = eng2.qr_begin(eng2.to_device(U), 0, r)
            plain = pivot_loop(eng2, st2, r)
            assert kinds.count('pool') >= 1 and sweeps <= plain and (sweeps < plain or r <= 2 * eng.qr_batch)
    else:
        assert set(kinds) <= {'refresh'} and sweeps == 1 + len(kinds)


@pytest.mark.parametrize('m,modes', [(128, ('number', 16)), (160, ('number', 40)), (128, ('variance', 99.0)), (96, ('number', 48))])
def test_top_r_eigen_route_matches_full_solve(m, modes, monkeypatch):
    """fit() above m = 96 takes only the r leading eigenvectors of the Gram matrix (dsytrd + dsterf + dstein + dormqr,
    sparse_sensing._eigvecs_top) instead of dsyevd's full decomposition (reference :272 computes all of them and :336
    slices): same rank, spectrum, basis (up to sign), sensors and field -- against the full solve on the same data and
    against the oracle."""
    import openmeasure_amd._eigen as ss
    from oracle import spr_oracle as orc
    X = _synth(400, 3, m, 60, 0.85, 1e-3, seed=m)
    select, n_modes = modes
    used = []
    real_top = ss._eigvecs_top

    def spy(fac, lam, r):
        V = real_top(fac, lam, r)
        used.append((r, V is not None))
        return V
    monkeypatch.setattr(ss, '_eigvecs_top', spy)
    a = SPR(X, 3, None, engine=NumpyEngine())
    a.fit(select_modes=select, n_modes=n_modes)
    assert used and used[-1][1] and used[-1][0] == a.r, 'the top-r route did not run'
    monkeypatch.setattr(ss, '_EIGH_TOP_MIN_M', 10 ** 9)            # the full dsyevd route on the same data
    b = SPR(X, 3, None, engine=NumpyEngine())
    b.fit(select_modes=select, n_modes=n_modes)
    assert a.r == b.r and 2 * a.r <= m
    np.testing.assert_allclose(a.S_ ** 2, b.S_ ** 2, rtol=0, atol=1e-12 * b.S_[0] ** 2)   # eigenvalues: dsterf vs dsyevd
    np.testing.assert_allclose(a.exp_variance_, b.exp_variance_, rtol=1e-12)
    sg = np.sign(np.sum(a.Ur * b.Ur, axis=0))
    assert np.abs(a.Ur * sg - b.Ur).max() <= 1e-9 * np.abs(b.Ur).max()
    # orthonormal to what the Gram route gives at this sigma_1/sigma_r (eps kappa^2), and no worse than the full solve
    oa, ob = np.abs(a.Ur.T @ a.Ur - np.eye(a.r)).max(), np.abs(b.Ur.T @ b.Ur - np.eye(b.r)).max()
    assert oa < 1e-8 and oa < 10 * ob + 1e-13
    a.optimal_placement(); b.optimal_placement()
    np.testing.assert_array_equal(a.sensors_, b.sensors_)
    st = orc.fit(X, 3, select_modes=select, n_modes=n_modes) if hasattr(orc, 'fit') else None
    if st is not None:
        assert st['r'] == a.r
        np.testing.assert_allclose(a.Sigma_r, st['Sigma_r'], rtol=1e-8)
        piv, _ = orc.qr_pivots(st['Ur'])
        np.testing.assert_array_equal(a.sensors_, piv)


def test_top_r_route_falls_back(monkeypatch):
    """r > m/2, a spectrum that needs the refinement pass, or vectors that fail the orthogonality check: dsyevd as before."""
    import openmeasure_amd._eigen as ss
    X = _synth(300, 2, 128, 100, 0.97, 1e-3, seed=3)
    calls = []
    real_top = ss._eigvecs_top
    monkeypatch.setattr(ss, '_eigvecs_top', lambda fac, lam, r: calls.append(r) or real_top(fac, lam, r))
    a = SPR(X, 2, None, engine=NumpyEngine())
    a.fit(select_modes='number', n_modes=100)                      # r > m/2: full solve, the top-r half never runs
    assert calls == [] and a.r == 100 and a.Ur.shape[1] == 100
    monkeypatch.setattr(ss, '_eigvecs_top', lambda fac, lam, r: None)   # a failed dstein / orthogonality check
    b = SPR(X, 2, None, engine=NumpyEngine())
    b.fit(select_modes='number', n_modes=12)
    assert b.r == 12 and np.abs(b.Ur.T @ b.Ur - np.eye(12)).max() < 1e-10


def _offset_data(axis_cnt, seed=5):
    """Fields whose centre dwarfs their fluctuation (pressure / temperature): row means 1e6 x the fluctuation and different from
    row to row (axis_cnt = 1); a block mean 1e7 x the block's spread (axis_cnt = None)."""
    rng = np.random.default_rng(seed)
    n_points, F, m = 600, 2, 24
    n = n_points * F
    fl = rng.standard_normal((n, 8)) @ ((0.7 ** np.arange(8))[:, None] * rng.standard_normal((8, m))) + 1e-3 * rng.standard_normal((n, m))
    if axis_cnt == 1:
        return 1e6 * (1.0 + rng.random((n, 1))) + fl, F
    return 1e7 + fl, F


@pytest.mark.parametrize('axis_cnt', [1, None])
def test_fit_chooses_the_precentred_projection_on_large_offsets(axis_cnt):
    """the safeguard has to compare the centre the projection's epilogue cancels with what is left AFTER the
    cancellation -- the rows' own fluctuation for row centring (the block std contains the spread of the row means and hid
    a ratio of 1e6 behind a 5), the block's spread for scalar centring (where the safeguard used to be switched off)."""
    X, F = _offset_data(axis_cnt)
    spr = SPR(X, F, None, engine=NumpyEngine())
    spr.fit(axis_cnt=axis_cnt, select_modes='number', n_modes=6)
    assert spr.precentered_ is True
    assert spr._pc_ratio.max() > 1e5
    Y, _ = _offset_data(axis_cnt)
    Y = Y - Y.mean(axis=1, keepdims=True) if axis_cnt == 1 else Y - 1e7            # the same fluctuation without the offset
    ref = SPR(np.ascontiguousarray(Y + 0.0), F, None, engine=NumpyEngine())
    ref.fit(axis_cnt=axis_cnt, select_modes='number', n_modes=6)
    assert ref.precentered_ is False


def test_rank_beyond_the_kernel_cap_is_a_value_error(monkeypatch):
    """r > SPR_MAX_R_WIDE retained modes: fit and reconstruct take them (:336 slices any r <= m); the placement and solve
    kernels do not -- a ValueError naming the cap, before any device work, instead of a failure inside the engine."""
    import openmeasure_amd.rom as ss
    monkeypatch.setattr(ss, 'SPR_MAX_R_WIDE', 6)
    rng = np.random.default_rng(4)
    X = rng.standard_normal((60, 12))
    spr = SPR(X, 2, rng.random((30, 3)), engine=NumpyEngine())
    spr.fit(select_modes='number', n_modes=8)
    assert spr.reconstruct(spr.Ar[0]).shape == (60, 1)
    for call in (spr.optimal_placement, lambda: spr.optimal_placement(calc_type='gem', n_sensors=3)):
        with pytest.raises(ValueError, match='exceed the 6 '):
            call()
    spr.fit(select_modes='number', n_modes=6)
    C = spr.optimal_placement()
    spr.train(C)
    spr.r = 8                                                       # a predict on a basis beyond the cap
    with pytest.raises(ValueError, match='exceed the 6 '):
        spr._solve([np.zeros((6, 3))])


def test_one_hot_rows_scalar_index_out_of_range():
    """C[i, j] with j outside [-n, n) raises like the dense ndarray it stands in for."""
    from openmeasure_amd.sparse_sensing import OneHotRows
    C = OneHotRows([3, 7], 10)
    dense = np.asarray(C)
    for j in (-10, -1, 0, 3, 9):
        assert C[0, j] == dense[0, j] and C[1, j] == dense[1, j]
    for j in (10, 12, -11):
        with pytest.raises(IndexError):
            dense[1, j]
        with pytest.raises(IndexError):
            C[1, j]


def _tridiag_case(kind, m, rng):
    if kind == 'random':
        return rng.standard_normal(m), rng.standard_normal(m - 1)
    if kind == 'decaying':                                  # the spectrum of a Gram matrix: eigenvalues over twelve decades
        lam = 10.0 ** (-12.0 * np.arange(m) / m)
        Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
        from scipy.linalg import lapack
        c, d, e, tau, info = lapack.dsytrd(np.asfortranarray((Q * lam) @ Q.T), lower=1)
        return d, e
    if kind == 'split':                                     # exact zeros on the off-diagonal: the matrix decouples
        d, e = rng.standard_normal(m), rng.standard_normal(m - 1)
        e[m // 3] = 0.0
        e[m // 2] = 0.0
        return d, e
    raise ValueError(kind)


@pytest.mark.parametrize('kind,m,r', [('random', 40, 40), ('random', 257, 64), ('decaying', 128, 48), ('split', 90, 30),
                                      ('random', 2, 2), ('random', 1, 1), ('random', 300, 7)])
def test_host_tridiagonal_vectors_vs_lapack(kind, m, r):
    """spr_host_tridiag_vectors (csrc/host_eig.hip): the inverse iterations of dstein for all requested eigenvalues side by
    side.  Residual and orthogonality against the tridiagonal matrix itself; vectors against numpy's eigh of it."""
    from openmeasure_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(m * 7 + r)
    d, e = _tridiag_case(kind, m, rng)
    d, e = np.ascontiguousarray(d, dtype=np.float64), np.ascontiguousarray(e, dtype=np.float64)
    T = np.diag(d) + (np.diag(e, 1) + np.diag(e, -1) if m > 1 else 0.0)
    lam, V = np.linalg.eigh(T)
    w = np.ascontiguousarray(lam[m - r:])
    Z = np.empty((m, r))
    assert lib.spr_host_tridiag_vectors(d.ctypes.data, e.ctypes.data if m > 1 else None, m, w.ctypes.data, r, Z.ctypes.data, 4) == 0
    scale = max(np.abs(lam).max(), 1e-300)
    assert np.abs(T @ Z - Z * w).max() <= 50 * np.finfo(float).eps * scale * np.sqrt(m)
    np.testing.assert_allclose(np.linalg.norm(Z, axis=0), 1.0, rtol=1e-14)
    gaps = np.diff(lam)
    if kind != 'split' and (m < 3 or gaps.min() > 1e-6 * scale):            # separated eigenvalues: the vectors themselves
        ref = V[:, m - r:]
        assert np.abs(np.abs(Z.T @ ref) - np.eye(r)).max() <= 1e-9
    # argument validation: no work on bad input
    assert lib.spr_host_tridiag_vectors(None, None, m, w.ctypes.data, r, Z.ctypes.data, 4) == -1
    assert lib.spr_host_tridiag_vectors(d.ctypes.data, e.ctypes.data if m > 1 else None, m, w.ctypes.data, m + 1, Z.ctypes.data, 4) == -1


def test_clustered_spectrum_takes_dstein(monkeypatch):
    """A Gram matrix with a (numerically) multiple retained eigenvalue: the batched inverse iteration cannot separate the
    cluster, _tridiag_vectors_batched declines (or its result fails the final check) and dstein / dsyevd take over --
    fit() still returns an orthonormal basis with the right spectrum."""
    import openmeasure_amd._eigen as ss
    rng = np.random.default_rng(8)
    m, n = 128, 3000
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    sig = np.r_[np.full(6, 5.0), 3.0, 3.0 * (1 + 1e-13), np.full(4, 1.0), 10.0 ** -np.linspace(1, 6, m - 12)]
    U, _ = np.linalg.qr(rng.standard_normal((n, m)))
    X = (U * sig) @ Q.T
    calls = []
    real = ss._tridiag_vectors_batched
    monkeypatch.setattr(ss, '_tridiag_vectors_batched', lambda d, e, w: calls.append(real(d, e, w)) or calls[-1])
    rom = SPR(np.ascontiguousarray(X), 1, None, engine=NumpyEngine())
    rom.fit(scale_type='none', select_modes='number', n_modes=12)
    assert calls and calls[-1] is None                       # multiple eigenvalues: declined
    assert np.abs(rom.Ur.T @ rom.Ur - np.eye(12)).max() < 1e-9
    Xc = X - X.mean(axis=1, keepdims=True)
    np.testing.assert_allclose(rom.Sigma_r, np.linalg.svd(Xc, compute_uv=False)[:12], rtol=1e-9)
