"""The reference's own known-answer unit tests for this path -- tests/test_rom.py (TestROM, 10 cases) and tests/test_spr.py
(TestSPR, 3 cases) of the reference, SURVEY.md 4(a) -- restated case by case, under the same names, on the product's class
mirror.  Every case runs twice: on the NumPy test double of the engine (CPU, the host logic alone) and on the HIP engine
through the C ABI (``-m gpu``).

What differs from the reference's assertions, and why:
  * the reference draws unseeded data; the cases here are seeded (three seeds each) so that a failure can be replayed;
  * where the reference asserts bit-equality with NumPy expressions (centring, scaling, X0) the device forms the same
    quantities with a different summation order (fused row sums / Chan-merged block statistics in f64): the tolerance is
    a few ulp, written next to each assertion;
  * ``decomposition`` is compared with ``np.linalg.svd`` up to the sign of every column and to 1e-9 (the Gram route cannot be
    bit-equal to LAPACK's dgesdd), and without the last mode: row-centred data of m = 5 snapshots has rank
    4, and the fifth singular pair is rounding noise in the reference as well."""
import numpy as np
import pytest

from openmeasure_amd.sparse_sensing import ROM, SPR

N_POINTS, N_FEATURES, M = 10, 2, 5             # the shape of the reference's setup_method (test_rom.py:7-14, test_spr.py:7-15)
ULP = np.finfo(float).eps

ENGINES = ['numpy', pytest.param('hip', marks=pytest.mark.gpu)]
_engines = {}


def _engine(kind):
    if kind not in _engines:
        if kind == 'numpy':
            from tests.numpy_engine import NumpyEngine
            _engines[kind] = NumpyEngine()
        else:
            from openmeasure_amd.engine import HipEngine
            _engines[kind] = HipEngine()
    return _engines[kind]


@pytest.fixture(params=[0, 1, 2], ids=lambda s: f'seed{s}')
def data(request):
    rng = np.random.default_rng(request.param)
    X = rng.random(size=(N_POINTS * N_FEATURES, M))
    xyz = rng.random(size=(N_POINTS, 3))
    return X, xyz


def _block(v, f):
    return v[f * N_POINTS:(f + 1) * N_POINTS]


def _feature_std(X):
    """np.std of every feature block, one value per row (the reference's expectation in test_scaling)"""
    out = np.zeros((X.shape[0], 1))
    for f in range(N_FEATURES):
        _block(out, f)[:] = np.std(_block(X, f))
    return out


def _same_up_to_sign(A, B, rtol):
    s = np.sign(np.sum(A * B, axis=0))
    np.testing.assert_allclose(A * s, B, rtol=rtol, atol=rtol * np.abs(B).max())


@pytest.mark.parametrize('kind', ENGINES)
class TestROM:
    def rom(self, data, kind):
        X, xyz = data
        return ROM(X, N_FEATURES, xyz, engine=_engine(kind))

    def test_centering_axis_one(self, data, kind):
        rom = self.rom(data, kind)
        rom.scale_data()
        np.testing.assert_allclose(rom.X_cnt, np.mean(rom.X, axis=1)[:, None], rtol=4 * ULP)        # ref: bit-equal

    def test_centering_axis_none(self, data, kind):
        rom = self.rom(data, kind)
        rom.scale_data(axis_cnt=None)
        want = np.zeros((rom.X.shape[0], 1))
        for f in range(N_FEATURES):
            _block(want, f)[:] = np.mean(_block(rom.X, f))
        np.testing.assert_allclose(rom.X_cnt, want, rtol=8 * ULP)                                    # ref: bit-equal

    def test_scaling(self, data, kind):
        rom = self.rom(data, kind)
        rom.scale_data()
        np.testing.assert_allclose(rom.X_scl, _feature_std(rom.X), rtol=16 * ULP)                    # ref: bit-equal

    def test_centering_and_scaling(self, data, kind):
        rom = self.rom(data, kind)
        X0 = rom.scale_data()
        want = (rom.X - np.mean(rom.X, axis=1)[:, None]) / _feature_std(rom.X)
        np.testing.assert_allclose(X0, want, rtol=64 * ULP, atol=16 * ULP)                           # ref: bit-equal

    def test_decomposition_svd(self, data, kind):
        rom = self.rom(data, kind)
        X0 = rom.scale_data()
        U, Sigma, Vt = np.linalg.svd(X0, full_matrices=False)
        A = (np.diag(Sigma) @ Vt).T
        Ur, Ar, _ = rom.decomposition(X0, n_modes=100)
        assert Ur.shape == U.shape and Ar.shape == A.shape
        _same_up_to_sign(Ur[:, :M - 1], U[:, :M - 1], 1e-9)                                          # ref: bit-equal to dgesdd
        _same_up_to_sign(Ar[:, :M - 1], A[:, :M - 1], 1e-9)

    def test_reduction_number(self, data, kind):
        rom = self.rom(data, kind)
        rom.decomposition(rom.scale_data(), select_modes='number', n_modes=M - 1)
        assert rom.r == M - 1

    def test_reduction_variance(self, data, kind):
        rom = self.rom(data, kind)
        rom.decomposition(rom.scale_data(), select_modes='variance', n_modes=100)
        assert rom.r == M

    def test_fit(self, data, kind):
        rom = self.rom(data, kind)
        X0 = rom.scale_data()
        _, Sigma, Vt = np.linalg.svd(X0, full_matrices=False)
        rom.fit(n_modes=100)
        np.testing.assert_allclose(rom.Sigma_r[:M - 1], Sigma[:M - 1], rtol=1e-7)                    # ref: assert_allclose default
        assert rom.Sigma_r[M - 1] <= 1e-7 * Sigma[0]                                                 # the null mode of row-centred data
        _same_up_to_sign(rom.Vr[:, :M - 1], Vt.T[:, :M - 1], 1e-7)

    def test_unscaling(self, data, kind):
        rom = self.rom(data, kind)
        X0 = rom.scale_data()
        rom.fit(n_modes=100)
        np.testing.assert_allclose(rom.unscale_data(X0[:, 0]), rom.X[:, 0])

    def test_reconstruction(self, data, kind):
        rom = self.rom(data, kind)
        rom.fit(n_modes=100)
        x_rec = rom.reconstruct(rom.Ar[0, :])
        assert x_rec.shape == (rom.X.shape[0], 1)
        np.testing.assert_allclose(x_rec, rom.X[:, [0]])


@pytest.mark.parametrize('kind', ENGINES)
class TestSPR:
    def spr(self, data, kind):
        X, xyz = data
        return SPR(X, N_FEATURES, xyz, engine=_engine(kind)), np.eye(X.shape[0])

    @staticmethod
    def measurement(spr, C):
        """column 0 of X seen through C, no uncertainty, feature index in the third column (test_spr.py:38-42)"""
        y = np.zeros((C.shape[0], 3))
        y[:, 0] = C @ spr.X[:, 0]
        for f in range(N_FEATURES):
            _block(y[:, 2], f)[:] = f
        return y

    def test_optimal_placement_qr(self, data, kind):
        spr, _ = self.spr(data, kind)
        spr.fit(n_modes=100)
        C_qr = spr.optimal_placement()
        assert C_qr.shape[0] == M
        assert C_qr.shape[1] == spr.X.shape[0]

    def test_scale_vector(self, data, kind):
        spr, C = self.spr(data, kind)
        spr.fit(n_modes=100)
        spr.train(C)
        y = self.measurement(spr, C)
        y0 = spr.scale_vector(y)
        want = np.zeros((C.shape[0], 2))
        want[:, 0] = (y[:, 0] - np.mean(spr.X, axis=1)) / _feature_std(spr.X)[:, 0]
        np.testing.assert_allclose(y0, want, atol=64 * ULP)

    def test_predict(self, data, kind):
        spr, C = self.spr(data, kind)
        spr.fit(n_modes=100)
        spr.train(C)
        a, _ = spr.predict(self.measurement(spr, C))
        x_pred = spr.reconstruct(a)
        np.testing.assert_allclose(x_pred, spr.X[:, [0]])
