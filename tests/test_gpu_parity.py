"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
  * the golden fixtures produced by the reference (tests/golden), and
  * the NumPy oracle (oracle/spr_oracle.py) on seeded inputs at sizes it finishes in seconds,
plus size-independent properties at larger sizes.
Tolerances: sensor indices exact and ordered; reconstructed fields <= 1e-6 relative
Frobenius (BASELINE.json north_star); intermediate quantities as stated inline."""
import os

import numpy as np
import pytest

from oracle import spr_oracle as orc
from tests.parity import (REL_FRO, align_signs, rel_fro, run_documented_idioms, run_f32_storage, run_fixture, run_gem_fixture, run_gpr_style,
                          run_conditioning_guard, run_gem_beyond_rank, run_pinv_fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from openmeasure_amd.engine import HipEngine
    return HipEngine()


def synth_host(n_points, F, m, k, rho, eps, seed):
    rng = np.random.default_rng(seed)
    n = n_points * F
    L = rng.standard_normal((n, k))
    R = (rho ** np.arange(k))[:, None] * rng.standard_normal((k, m))
    X = L @ R + eps * rng.standard_normal((n, m))
    for f in range(F):
        X[f * n_points:(f + 1) * n_points] = (f + 1) * X[f * n_points:(f + 1) * n_points] + 10.0 * f
    return np.ascontiguousarray(X)


def test_native_library_is_loaded(eng):
    import os
    maps = open(f'/proc/{os.getpid()}/maps').read()
    assert 'libspr_hip.so' in maps
    from openmeasure_amd import _lib
    assert eng.lib.spr_abi_version() == _lib.SPR_ABI_VERSION == 5


def test_golden_fixture(golden, eng):
    run_fixture(golden, eng)


@pytest.mark.parametrize('n_points,F,m,r', [(400, 3, 12, 4), (3000, 4, 64, 32), (1500, 9, 41, 14), (700, 3, 256, 64),
                                            (777, 2, 255, 33), (500, 5, 300, 40), (1000, 16, 512, 128)])
def test_f32_storage_end_to_end(eng, n_points, F, m, r):
    """f32-stored shard and basis (BASELINE config 5's storage), f64 arithmetic: _x32 / _u32 entry points."""
    run_f32_storage(eng, n_points, F, m, r, 33, synth_host)


def test_f32_storage_kernels_vs_f64_twins(eng):
    """Kernel by kernel: the f32-storage entry points on X32 equal the f64 ones on X32 widened (same arithmetic),
    up to the one rounding of the stored basis."""
    import torch
    rng = np.random.default_rng(7)
    n_points, F, m, r = 2001, 3, 96, 24
    X32 = (rng.standard_normal((n_points * F, m)) * 3 + 1).astype(np.float32)
    Xs, Xw = eng.to_device(X32, dtype=torch.float32), eng.to_device(X32.astype(np.float64))
    ms, fs, gs = eng.stats_gram(Xs, 0, n_points, F)
    mw, fw, gw = eng.stats_gram(Xw, 0, n_points, F)
    np.testing.assert_array_equal(eng.to_host(ms), eng.to_host(mw))
    np.testing.assert_array_equal(eng.to_host(gs), eng.to_host(gw))
    np.testing.assert_array_equal(eng.to_host(fs), eng.to_host(fw))
    np.testing.assert_array_equal(eng.to_host(eng.feature_minmax(Xs, 0, n_points, F)), eng.to_host(eng.feature_minmax(Xw, 0, n_points, F)))
    np.testing.assert_array_equal(eng.to_host(eng.colsums(Xs, 0, n_points, F, ms)), eng.to_host(eng.colsums(Xw, 0, n_points, F, mw)))
    W = eng.to_device(rng.standard_normal((m, r)))
    inv = eng.to_device(np.ones(F))
    Us = eng.project(Xs, 0, n_points, F, inv, W, rowmean=ms, basis_dtype=torch.float32)
    Uw = eng.project(Xw, 0, n_points, F, inv, W, rowmean=mw)
    assert Us.dtype == torch.float32 and Uw.dtype == torch.float64
    np.testing.assert_array_equal(eng.to_host(Us), eng.to_host(Uw).astype(np.float32))
    Ud = eng.project(Xs, 0, n_points, F, inv, W, rowmean=ms)            # default: float64 basis of the f32 shard
    assert Ud.dtype == torch.float64
    np.testing.assert_array_equal(eng.to_host(Ud), eng.to_host(Uw))
    Uq = eng.to_device(eng.to_host(Us).astype(np.float64))            # the stored basis, widened
    a = eng.to_device(rng.standard_normal((5, r)))
    sc = eng.to_device(np.array([1.0, 2.0, 0.5]))
    np.testing.assert_array_equal(eng.to_host(eng.reconstruct(Us, 0, n_points, F, ms, sc, a)),
                                  eng.to_host(eng.reconstruct(Uq, 0, n_points, F, ms, sc, a)))
    s1, s2 = eng.qr_begin(Us, 0, 4), eng.qr_begin(Uq, 0, 4)
    np.testing.assert_array_equal(eng.to_host(s1['nrm']), eng.to_host(s2['nrm']))
    np.testing.assert_array_equal(eng.to_host(s1['rec']), eng.to_host(s2['rec']))
    for st in (s1, s2):
        eng.qr_step(st, 0, st['rec'][None], st['tau'][None], True)
        eng.qr_refresh(st, 0, 1)
    np.testing.assert_array_equal(eng.to_host(s1['nrm']), eng.to_host(s2['nrm']))
    np.testing.assert_array_equal(eng.to_host(s1['piv'][:1]), eng.to_host(s2['piv'][:1]))


def test_gem_fixture(golden_gem, eng):                    # calc_type='gem', picks pinned by the reference
    run_gem_fixture(golden_gem, eng)


@pytest.mark.parametrize('n_points,F,r,s,d_min,masked,seed', [(20000, 3, 16, 15, 0.05, False, 1), (5000, 4, 32, 20, 0.0, True, 2),
                                                             (100000, 2, 8, 7, 0.02, True, 3), (3000, 1, 64, 40, 0.1, False, 4)])
def test_gem_vs_oracle(eng, n_points, F, r, s, d_min, masked, seed):
    """GEM on random bases large enough for several candidate-set batches, against the oracle's literal formulas."""
    from openmeasure_amd.sparse_sensing import SPR
    rng = np.random.default_rng(seed)
    n = n_points * F
    Ur = rng.standard_normal((n, r)) * (0.5 + rng.random((n, 1))) / np.sqrt(n)
    xyz = rng.random((n_points, 3))
    mask = rng.random(n) < 0.6 if masked else None
    want, lead = orc.gem_pivots(Ur, s, xyz, F, mask, d_min)
    spr = SPR(np.zeros((n, r + 2)), F, xyz, engine=eng)
    spr.fit(basis=(Ur, np.eye(r + 2, r)))
    C = spr.optimal_placement(calc_type='gem', n_sensors=s, mask=mask, d_min=d_min)
    assert C.shape == (s, n)
    k = int(np.argmax(lead < 1e-9)) if (lead < 1e-9).any() else s     # compare up to the first numerical tie
    np.testing.assert_array_equal(spr.sensors_[:k], want[:k])
    assert k >= min(s, 5)
    if d_min > 0:                                          # no two sensors closer than d_min
        p = xyz[spr.sensors_ % n_points]
        d = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(s)
        assert d.min() >= d_min
    if mask is not None:
        assert mask[spr.sensors_].all()


@pytest.mark.parametrize('foreign', [False, True])
def test_gpr_style_subclass(golden, eng, foreign):       # gpr.py:379-402: ROM as the base class of GPR
    run_gpr_style(golden, eng, foreign_basis=foreign)


@pytest.mark.parametrize('n_points,F,m,seed', [(5000, 3, 64, 0), (1001, 2, 7, 1), (257, 5, 256, 2), (40000, 1, 1, 3)])
def test_median_scaling_vs_numpy(eng, n_points, F, m, seed):      # :140-141, radix selection (csrc/select.hip)
    from openmeasure_amd.sparse_sensing import ROM
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n_points * F, m)) * np.repeat(10.0 ** rng.integers(-3, 4, F), n_points)[:, None]
    X[::3] = np.round(X[::3], 1)                                    # many exact ties, both signs
    X += np.repeat(rng.integers(-1, 2, F), n_points)[:, None] * 2.5
    rom = ROM(X, F, None, engine=eng)
    rom.scale_data('median')
    want = np.array([np.median(X[f * n_points:(f + 1) * n_points]) for f in range(F)])
    np.testing.assert_array_equal(rom._scl_f, want)
    # per-shard histograms add up to the histogram of the whole block (what the all-reduce relies on)
    import torch
    Xd = eng.to_device(X)
    pre = eng.to_device(np.zeros((F, 2), dtype=np.int64), dtype=torch.int64)
    whole = eng.to_host(eng.feature_digit_hist(Xd, 0, n_points, F, pre, 51, 13, False))
    cut = (n_points * F) // 3 + 1
    parts = (eng.to_host(eng.feature_digit_hist(Xd[:cut], 0, n_points, F, pre, 51, 13, False))
             + eng.to_host(eng.feature_digit_hist(Xd[cut:], cut, n_points, F, pre, 51, 13, False)))
    np.testing.assert_array_equal(whole, parts)
    assert whole[:, 0].sum(axis=1).tolist() == [n_points * m] * F


def test_upload_download_staging(eng):
    """to_device / to_host go through pinned staging (spr_upload_bytes kernel, pinned D2H): values, dtypes, ring reuse."""
    import torch
    rng = np.random.default_rng(5)
    keep = []
    for i in range(3 * eng._STAGE_SLOTS):                 # more uploads than ring slots, all kept alive
        a = rng.standard_normal((i + 1, 7))
        keep.append((a, eng.to_device(a)))
    for a, t in keep:
        np.testing.assert_array_equal(eng.to_host(t), a)
    idx = np.arange(13, dtype=np.int32)
    t = eng.to_device(idx, dtype=torch.int64)
    assert t.dtype == torch.int64 and eng.to_host(t).tolist() == list(range(13))
    odd = eng.to_device(np.arange(5, dtype=np.uint8), dtype=torch.uint8)      # 5 bytes: not a multiple of 8
    assert eng.to_host(odd).tolist() == [0, 1, 2, 3, 4]
    big = rng.standard_normal((300, 1000))                # 2.4 MB: above the upload staging size
    np.testing.assert_array_equal(eng.to_host(eng.to_device(big)), big)
    view = eng.to_device(big)[::2, 1:50]                  # non-contiguous download
    np.testing.assert_array_equal(eng.to_host(view), big[::2, 1:50])


# ---- kernel-level parity against the oracle, shapes chosen to hit every template family ----
@pytest.mark.parametrize('n_points,F,m', [
    (10, 2, 5), (333, 3, 7), (257, 1, 16), (1000, 3, 33), (4099, 2, 48), (700, 4, 64),
    (513, 2, 80), (1200, 3, 128), (640, 2, 190), (900, 9, 256), (31, 1, 256),
    (400, 2, 257), (333, 3, 300), (500, 2, 384), (450, 4, 511), (260, 16, 512),
    (300, 2, 513), (260, 3, 600), (200, 2, 1024), (150, 1, 1100),      # slice pairs (csrc/gram_wide.hip, any m)
])
def test_stats_gram_vs_oracle(eng, n_points, F, m):
    X = synth_host(n_points, F, m, min(m, 12), 0.8, 1e-3, 1000 + m)
    X_cnt, X_scl, X0 = orc.scale_data_std(X, F)
    rowmean, fstats, gram = eng.stats_gram(eng.to_device(X), 0, n_points, F)
    rowmean, fstats, gram = eng.to_host(rowmean), eng.to_host(fstats), eng.to_host(gram)
    np.testing.assert_allclose(rowmean, X_cnt[:, 0], rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(fstats[:, 0], n_points)
    var = (np.trace(gram, axis1=1, axis2=2) + m * fstats[:, 2]) / (n_points * m)
    np.testing.assert_allclose(np.sqrt(var), X_scl[::n_points, 0], rtol=1e-12)
    G = np.sum(gram / var[:, None, None], axis=0)
    Gref = X0.T @ X0
    assert np.abs(G - Gref).max() <= 1e-12 * np.abs(Gref).max()
    np.testing.assert_array_equal(G, G.T)               # mirrored exactly


@pytest.mark.parametrize('n_points,F,m,r', [
    (10, 2, 5, 4), (333, 3, 7, 5), (1000, 3, 33, 17), (700, 4, 64, 32), (513, 2, 80, 40),
    (1200, 3, 128, 64), (900, 9, 256, 64), (300, 2, 256, 128), (2000, 1, 48, 1),
    (333, 3, 300, 33), (400, 2, 384, 64), (260, 16, 512, 128),
])
def test_project_vs_oracle(eng, n_points, F, m, r):
    X = synth_host(n_points, F, m, min(m, 2 * r), 0.9, 1e-3, 2000 + m + r)
    X_cnt, X_scl, X0 = orc.scale_data_std(X, F)
    rng = np.random.default_rng(1)
    W = rng.standard_normal((m, r))
    inv = eng.to_device(1.0 / X_scl[::n_points, 0])
    U = eng.to_host(eng.project(eng.to_device(X), 0, n_points, F, inv, eng.to_device(W),
                                rowmean=eng.to_device(X_cnt[:, 0])))
    ref = X0 @ W
    assert np.abs(U - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize('n_points,F,m,r,row0_cells,f32', [
    (3000, 3, 256, 64, 0, False), (2000, 4, 256, 64, 1500, False), (2500, 3, 256, 33, 0, False), (1700, 5, 256, 17, 300, False),
    (3000, 2, 128, 64, 0, False), (2100, 3, 192, 48, 0, False), (1500, 3, 128, 16, 300, False), (4099, 1, 256, 64, 0, False),
    (3000, 3, 256, 64, 0, True), (2222, 3, 128, 30, 111, True),
    (3000, 3, 64, 32, 0, False), (2100, 4, 64, 17, 700, False), (4099, 1, 64, 64, 0, False), (2500, 2, 64, 9, 0, True),
])
def test_project_w_stationary_vs_oracle(eng, n_points, F, m, r, row0_cells, f32):
    """The W-stationary projection kernel (csrc/project_ws.hip: n >= 4096 rows, m in {64, 128, 192, 256} packed, r <= 64):
    full and ragged tails, r not a multiple of 16, shards starting inside a feature (row0 != 0), f32 storage, and
    the un-centred form -- against ((X - mean) W) / X_scl formed in NumPy."""
    import torch
    X = synth_host(n_points, F, m, min(m, 2 * r), 0.9, 1e-3, 3000 + m + r)
    if f32:
        X = X.astype(np.float32).astype(np.float64)
    X_cnt, X_scl, X0 = orc.scale_data_std(X, F)
    rng = np.random.default_rng(2)
    W = rng.standard_normal((m, r))
    row0 = row0_cells                                           # local block = global rows [row0, n)
    Xl = X[row0:]
    assert Xl.shape[0] >= 4096
    inv = eng.to_device(1.0 / X_scl[::n_points, 0])
    Xd = eng.to_device(Xl.astype(np.float32), dtype=torch.float32) if f32 else eng.to_device(Xl)
    bd = torch.float32 if f32 else None                         # f32 basis = storage option: one rounding of the f64 result
    U = eng.to_host(eng.project(Xd, row0, n_points, F, inv, eng.to_device(W), rowmean=eng.to_device(X_cnt[row0:, 0]),
                                basis_dtype=bd))
    ref = X0[row0:] @ W
    tol = 2e-7 if f32 else 1e-12
    assert U.shape == ref.shape and np.abs(U - ref).max() <= tol * np.abs(ref).max()
    if f32:                                                     # default for an f32 shard: the reference's float64 basis
        U64 = eng.to_host(eng.project(Xd, row0, n_points, F, inv, eng.to_device(W), rowmean=eng.to_device(X_cnt[row0:, 0])))
        assert U64.dtype == np.float64 and np.abs(U64 - ref).max() <= 1e-12 * np.abs(ref).max()
    if not f32:
        ones = eng.to_device(np.ones(F))
        U0 = eng.to_host(eng.project(Xd, row0, n_points, F, ones, eng.to_device(W), center=False))
        ref0 = Xl @ W
        assert np.abs(U0 - ref0).max() <= 1e-12 * np.abs(ref0).max()


@pytest.mark.parametrize('seed', range(16))
def test_project_w_stationary_random_shapes(eng, seed):
    """Seeded random shapes through the W-stationary projection kernel: rows from 4096 to ~40k (ragged 16-row and
    128-row tails), 1-6 features with the shard starting and ending inside features, r from 1 to 64, m in
    {64, 128, 192, 256}, f64 and f32 storage, centred and not -- against NumPy."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.choice([128, 192, 256])) if seed < 12 else 64
    r = int(rng.integers(1, 65))
    F = int(rng.integers(1, 7))
    n_points = int(rng.integers(4096 // F + 400, 40000 // F))
    n = n_points * F
    row0 = int(rng.integers(0, n - 4096 - 1)) if seed % 3 else 0
    n_loc = int(rng.integers(4096, n - row0 + 1))
    f32 = bool(seed % 4 == 3)
    X = rng.standard_normal((n_loc, m)) * 3.0 + rng.standard_normal((n_loc, 1)) * 10.0
    if f32:
        X = X.astype(np.float32).astype(np.float64)
    mu = X.mean(axis=1)
    scl = 0.5 + rng.random(F)
    feat = (row0 + np.arange(n_loc)) // n_points
    W = rng.standard_normal((m, r))
    Xd = eng.to_device(X.astype(np.float32), dtype=torch.float32) if f32 else eng.to_device(X)
    U = eng.to_host(eng.project(Xd, row0, n_points, F, eng.to_device(1.0 / scl), eng.to_device(W), rowmean=eng.to_device(mu),
                                basis_dtype=torch.float32 if f32 else None))
    ref = ((X - mu[:, None]) @ W) / scl[feat][:, None]
    tol = 2e-7 if f32 else 1e-12
    assert U.shape == ref.shape
    assert np.abs(U - ref).max() <= tol * np.abs(ref).max(), (m, r, F, n_points, row0, n_loc, f32)


@pytest.mark.parametrize('seed', range(20))
def test_project_stream_random_shapes(eng, seed):
    """The streamed-W projection kernel (csrc/project_stream.hip: any m, r <= 128 per column group): m from 257 to 1100
    incl. widths that are not multiples of 4 / 32 (scalar loads, padded chunks), r from 1 to 300 (column groups of 128),
    ragged 16- and 256-row tails, shards starting inside a feature, f64 and f32 storage with f64 and f32 bases, centring
    in the epilogue, in registers (precenter) and none -- against ((X - mean) W) / X_scl formed in NumPy."""
    import torch
    rng = np.random.default_rng(5000 + seed)
    m = int([512, 300, 260, 384, 513, 600, 1024, 257, 640, 1100][seed % 10])
    r = int([128, 40, 16, 100, 33, 130, 64, 1, 300, 7][(seed // 2) % 10])
    F = int(rng.integers(1, 5))
    n_points = int(rng.integers(1200, 6000) // F + 1)
    n = n_points * F
    row0 = int(rng.integers(0, n // 3)) if seed % 3 else 0
    n_loc = int(rng.integers(max(1, (n - row0) // 2), n - row0 + 1))
    f32 = seed % 4 == 1
    f32_basis = f32 and seed % 8 == 1
    X = rng.standard_normal((n_loc, m)) * 3.0 + rng.standard_normal((n_loc, 1)) * 10.0
    if f32:
        X = X.astype(np.float32).astype(np.float64)
    mu = X.mean(axis=1)
    scl = 0.5 + rng.random(F)
    feat = (row0 + np.arange(n_loc)) // n_points
    W = rng.standard_normal((m, r))
    Xd = eng.to_device(X.astype(np.float32), dtype=torch.float32) if f32 else eng.to_device(X)
    if seed % 5 == 4:                                           # a row stride that is not the width (and odd alignment)
        pad = torch.empty((n_loc, m + 3), dtype=Xd.dtype, device=Xd.device)
        pad[:, :m] = Xd
        Xd = pad[:, :m]
    bd = torch.float32 if f32_basis else None
    tol = 2e-7 if f32_basis else 1e-12
    ref = ((X - mu[:, None]) @ W) / scl[feat][:, None]
    for pre in (False, True):
        U = eng.to_host(eng.project(Xd, row0, n_points, F, eng.to_device(1.0 / scl), eng.to_device(W),
                                    rowmean=eng.to_device(mu), basis_dtype=bd, precenter=pre))
        assert U.shape == ref.shape and U.dtype == (np.float32 if f32_basis else np.float64)
        assert np.abs(U - ref).max() <= tol * np.abs(ref).max(), (m, r, F, n_points, row0, n_loc, f32, pre)
    U0 = eng.to_host(eng.project(Xd, row0, n_points, F, eng.to_device(np.ones(F)), eng.to_device(W), center=False,
                                 basis_dtype=bd))
    ref0 = X @ W
    assert np.abs(U0 - ref0).max() <= tol * np.abs(ref0).max(), (m, r, 'uncentred')


@pytest.mark.parametrize('n_points,F,m,r,f32,f32_basis,pre', [
    (9000, 3, 256, 64, False, False, False),     # W-stationary kernel (config 3/4 shape)
    (5000, 2, 128, 30, False, False, False),     # W-stationary, padded column tile
    (4100, 2, 192, 64, True, True, False),       # W-stationary, f32 shard and basis: norms of the ROUNDED rows
    (4100, 2, 192, 33, True, False, False),      # W-stationary, f32 shard, f64 basis
    (1000, 2, 256, 64, False, False, False),     # fewer than 4096 rows: not W-stationary -> streamed-W kernel
    (3000, 4, 64, 32, False, False, False),      # config-2 shape: streamed-W kernel
    (1500, 9, 41, 14, False, False, False),      # config-1 shape: odd m, scalar loads
    (2000, 16, 512, 128, True, True, False),     # config-5 shape
    (2500, 2, 300, 40, False, False, True),      # pre-centred operand
    (999, 3, 600, 128, True, False, False)])
def test_project_row_norms(eng, n_points, F, m, r, f32, f32_basis, pre):
    """project(norms=...) (spr_project_norms_* / spr_project_stream_norms_*): the vector holds the squared norms of the
    rows of Ur as STORED -- for an f32 basis of the rounded values, not of the f64 accumulators --, the basis itself is
    bitwise the one the plain call writes; ragged last block, shard starting inside a feature."""
    import torch
    rng = np.random.default_rng(m * 1000 + r)
    n = n_points * F
    row0 = n // 5
    n_loc = n - row0 - 7
    X = rng.standard_normal((n_loc, m)) * 2.0 + rng.standard_normal((n_loc, 1)) * 5.0
    if f32:
        X = X.astype(np.float32).astype(np.float64)
    Xd = eng.to_device(X.astype(np.float32), dtype=torch.float32) if f32 else eng.to_device(X)
    mu = eng.to_device(X.mean(axis=1))
    isc = eng.to_device(1.0 / (0.5 + rng.random(F)))
    W = eng.to_device(rng.standard_normal((m, r)) / np.sqrt(m))
    bd = torch.float32 if f32_basis else None
    nrm = eng.empty((n_loc,))
    nrm.fill_(-7.0)
    U = eng.project(Xd, row0, n_points, F, isc, W, rowmean=mu, basis_dtype=bd, precenter=pre, norms=nrm)
    U_plain = eng.project(Xd, row0, n_points, F, isc, W, rowmean=mu, basis_dtype=bd, precenter=pre)
    if not (eng.lib.spr_project_norms_supported(m, r, n_loc, Xd.stride(0), Xd.data_ptr(), int(f32)) == 0 and m <= 256
            and not pre):
        # same kernel with and without the norm epilogue: bitwise the same basis (a shape that only takes the
        # streamed-W kernel BECAUSE norms were asked for is compared to the general kernel's result instead)
        assert torch.equal(U, U_plain)
    else:
        assert float((U.double() - U_plain.double()).abs().max()) <= 1e-12 * float(U_plain.double().abs().max())
    ref = (U.double() ** 2).sum(dim=1)
    got = eng.to_host(nrm)
    np.testing.assert_allclose(got, eng.to_host(ref), rtol=2e-15 * max(r, 8), atol=0.0)


@pytest.mark.parametrize('n_points,F,m,r,f32_basis', [(20000, 4, 256, 64, False), (20000, 4, 64, 32, False),
                                                    (18362, 9, 41, 14, False), (6000, 16, 512, 128, True),
                                                    (3000, 2, 300, 100, False)])
def test_placement_from_fused_norms(eng, n_points, F, m, r, f32_basis):
    """ROM.placement_norms: fit() leaves the squared row norms, optimal_placement() starts from them -- one pass over the
    basis fewer, the same ordered sensors and gaps as the plain route (and as the oracle), also through a mask (which
    zeroes rows: the stored norms are dropped), a re-fit and an assigned basis."""
    import torch
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
    rho = 10 ** (-3 / (r - 1))
    X = synth_host(n_points, F, m, min(m, 2 * r), rho, 1e-3, 5 + m)
    if f32_basis:
        Xin = DeviceMatrix(eng.to_device(X.astype(np.float32), dtype=torch.float32), basis='f32')
    else:
        Xin = X
    plain = SPR(Xin, F, None, engine=eng)
    plain.placement_norms = False
    plain.fit(select_modes='number', n_modes=r)
    plain.optimal_placement()
    assert plain.placement_from_norms_ is False
    auto = SPR(Xin, F, None, engine=eng)                       # default: norms where the shape's own kernel writes them
    auto.fit(select_modes='number', n_modes=r)
    auto.optimal_placement()
    assert auto.placement_from_norms_ is (m > 256 or (m, r) in ((256, 64), (64, 32)))     # W-stationary / streamed-W shapes
    np.testing.assert_array_equal(auto.sensors_, plain.sensors_)
    fused = SPR(Xin, F, None, engine=eng)
    fused.placement_norms = True
    fused.fit(select_modes='number', n_modes=r)
    fused.optimal_placement()
    assert fused.placement_from_norms_ is True
    np.testing.assert_array_equal(fused.sensors_, plain.sensors_)
    np.testing.assert_allclose(fused.pivot_gap_, plain.pivot_gap_, rtol=1e-9, atol=1e-14)
    assert fused.pivot_sweeps_ == plain.pivot_sweeps_ - 1
    if not f32_basis:
        np.testing.assert_array_equal(fused.sensors_, orc.qr_pivots(orc.fit(X, F, 'number', r)['Ur'])[0])
    fused.optimal_placement()                                  # the stored vector is not consumed
    assert fused.placement_from_norms_ is True
    np.testing.assert_array_equal(fused.sensors_, plain.sensors_)
    mask = np.ones(X.shape[0], dtype=bool)
    mask[plain.sensors_[:3]] = False
    fused.optimal_placement(mask=mask)                         # rows zeroed in place (:737-738): norms no longer valid
    assert fused.placement_from_norms_ is False
    plain.optimal_placement(mask=mask)
    np.testing.assert_array_equal(fused.sensors_, plain.sensors_)
    fused.fit(select_modes='number', n_modes=r)                # a new fit leaves new norms
    fused.optimal_placement()
    assert fused.placement_from_norms_ is True
    fused.placement_norms = False
    fused.fit(select_modes='number', n_modes=max(r - 1, 1))
    fused.optimal_placement()
    assert fused.placement_from_norms_ is False and len(fused.sensors_) == max(r - 1, 1)


def test_project_precentred_large_means(eng):
    """Row means 1e6 times the fluctuation (pressure / temperature fields): removing the mean in the epilogue,
    x.W - mean (1^T W), loses 6 digits to cancellation; centre mode 2 of the streamed-W kernel subtracts it from the
    operand first -- the reference's own order (:169) -- and keeps f64 accuracy.  Both against longdouble NumPy."""
    rng = np.random.default_rng(77)
    n_points, F, m, r = 5000, 2, 256, 64
    n = n_points * F
    fl = rng.standard_normal((n, m))
    X = 1e6 * (1.0 + rng.random((n, 1))) + fl
    mu = X.mean(axis=1)
    W = rng.standard_normal((m, r))
    ref = np.asarray((X.astype(np.longdouble) - mu[:, None].astype(np.longdouble)) @ W.astype(np.longdouble), dtype=np.float64)
    args = (eng.to_device(X), 0, n_points, F, eng.to_device(np.ones(F)), eng.to_device(W))
    epi = eng.to_host(eng.project(*args, rowmean=eng.to_device(mu)))
    pre = eng.to_host(eng.project(*args, rowmean=eng.to_device(mu), precenter=True))
    e_epi, e_pre = np.abs(epi - ref).max() / np.abs(ref).max(), np.abs(pre - ref).max() / np.abs(ref).max()
    assert e_pre <= 1e-13, e_pre
    assert 1e-12 < e_epi < 1e-7, e_epi                         # the cancellation the pre-centred form avoids


@pytest.mark.parametrize('n,r,n_p', [(20, 5, 1), (999, 5, 3), (4096, 32, 1), (5000, 64, 5), (3001, 128, 2), (777, 1, 1), (1234, 14, 2),
                                     (2000, 129, 2), (1500, 300, 3), (900, 513, 1), (700, 1024, 2)])
def test_reconstruct_vs_oracle(eng, n, r, n_p):
    rng = np.random.default_rng(n + r)
    F = 1 if n % 3 else 3
    n_points = n // F
    Ur = rng.standard_normal((n, r))
    X_cnt = rng.standard_normal((n, 1)) * 10
    scl_f = 1.0 + rng.random(F)
    X_scl = np.repeat(scl_f, n_points)[:, None]
    A = rng.standard_normal((n_p, r))
    ref = orc.reconstruct(A, Ur, X_cnt, X_scl)
    Ud = eng.to_device(np.pad(Ur, ((0, 0), (0, r & 1))))[:, :r]
    out = eng.to_host(eng.reconstruct(Ud, 0, n_points, F, eng.to_device(X_cnt[:, 0]), eng.to_device(scl_f),
                                      eng.to_device(A))).T
    assert out.shape == (n, n_p)
    assert rel_fro(out, ref) <= 1e-14


def _full_path(eng, X, F, r, engine_kwargs=None):
    from openmeasure_amd.sparse_sensing import SPR
    n, m = X.shape
    n_points = n // F
    spr = SPR(X, F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    C = spr.optimal_placement()
    rng = np.random.default_rng(99)
    w = rng.standard_normal(m) / np.sqrt(m)
    xt = X @ w + X.mean(axis=1) * (1 - w.sum())          # held-out state in the span of the data

    def y_fn(piv):
        y = np.zeros((len(piv), 3))
        y[:, 0] = xt[piv]
        y[:, 2] = piv // n_points
        return y
    ref = orc.fit_place_train_predict_reconstruct(X, F, r, y_fn)
    spr.train(C)
    a, _ = spr.predict(y_fn(spr.sensors_))
    xr = spr.reconstruct(a)
    return spr, ref, xr


@pytest.mark.parametrize('n_points,F,m,r', [(18362, 9, 41, 14), (20000, 4, 64, 32), (6000, 3, 128, 16),
                                            (3000, 16, 512, 128), (2500, 2, 320, 24)])
def test_end_to_end_vs_oracle(eng, n_points, F, m, r):
    """config-1 shape (18 362 cells x 9 features x 41 snapshots, 14 sensors) and two others:
    ordered sensor indices exact, reconstructed field within 1e-6 rel-Frobenius."""
    rho = 10 ** (-3 / (r - 1))
    X = synth_host(n_points, F, m, min(m, 2 * r), rho, 1e-3, 31 + m)
    spr, ref, xr = _full_path(eng, X, F, r)
    np.testing.assert_allclose(spr.Sigma_r, ref['Sigma_r'], rtol=1e-8)
    np.testing.assert_array_equal(spr.sensors_, ref['piv'])
    assert spr.pivot_gap_.min() > 1e-9
    assert rel_fro(xr, ref['X_rec']) <= REL_FRO
    sg = align_signs(spr.Ur, ref['Ur'])
    assert np.abs(spr.Ur * sg - ref['Ur']).max() <= 1e-9


@pytest.mark.parametrize('seed', range(int(os.environ.get('SPR_TEST_SHAPE_SEEDS', '48'))))
def test_random_shapes_end_to_end(eng, seed):
    """48 seeded random shapes (SPR_TEST_SHAPE_SEEDS=<k> for a soak: tools/r05_soak.sh) -- 3 ... 4 000 cells, 1 ... 6 features, 2 ... 330 snapshots (narrow, 256-wide and wide Gram
    routes; every projection, placement and solve template the widths select), 1 ... 70 modes -- through fit -> placement ->
    train -> predict -> reconstruct against the oracle: spectrum to 1e-8, retained subspace to 1e-8, ordered sensors exact
    wherever the oracle's own pivot margin is above rounding, field within 1e-6 rel-Frobenius (north_star)."""
    rng = np.random.default_rng(1000 + seed)
    F = int(rng.integers(1, 7))
    m = int(rng.integers(2, 40)) if seed % 3 == 0 else int(rng.integers(40, 331))
    n_points = max(int(rng.integers(3, 4001)), (m + 2 + F - 1) // F)       # tall matrix: n >= m + 2
    r = int(rng.integers(1, min(m - 1, 70) + 1))
    rho = 10 ** (-3 / (r - 1)) if r > 1 else 0.5
    X = synth_host(n_points, F, m, min(m, 2 * r), rho, 1e-3, 5000 + seed)
    spr, ref, xr = _full_path(eng, X, F, r)
    assert spr.r == r == ref['Ur'].shape[1]
    np.testing.assert_allclose(spr.Sigma_r, ref['Sigma_r'], rtol=1e-8)
    Q = spr.Ur.T @ ref['Ur']                                                # same subspace <=> Q orthogonal
    assert np.abs(Q.T @ Q - np.eye(r)).max() <= 1e-8, (n_points, F, m, r)
    gaps = spr.pivot_gap_
    safe = len(gaps) if gaps.min() > 1e-9 else int(np.argmax(gaps <= 1e-9))  # steps before the first near-tie
    np.testing.assert_array_equal(spr.sensors_[:safe], ref['piv'][:safe], err_msg=str((n_points, F, m, r)))
    if safe == len(gaps):
        assert rel_fro(xr, ref['X_rec']) <= REL_FRO, (n_points, F, m, r)


@pytest.mark.parametrize('device_matrix,f32', [(False, False), (True, False), (True, True)])
def test_pickle_round_trip_on_the_device(eng, device_matrix, f32):
    """pickle.dumps of a fitted, trained object downloads what lives in HBM (a DeviceMatrix becomes a host ndarray of its
    values, an f32 basis stays f32); the loaded object creates its engine and uploads its state on first use and gives the
    same answers bit for bit -- also after a second fit."""
    import pickle
    import torch
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
    X = synth_host(3000, 3, 48, 24, 0.7, 1e-3, 11)
    if f32:
        X = X.astype(np.float32)
    Xin = DeviceMatrix(eng.to_device(X, dtype=torch.float32 if f32 else None), basis='f32' if f32 else None) if device_matrix else X
    spr = SPR(Xin, 3, None, engine=eng)
    spr.fit(select_modes='number', n_modes=12)
    C = spr.optimal_placement()
    spr.train(C)
    y = np.zeros((12, 3))
    y[:, 0] = X[spr.sensors_, 5]
    y[:, 2] = spr.sensors_ // 3000
    a0, _ = spr.predict(y)
    x0 = spr.reconstruct(a0)
    clone = pickle.loads(pickle.dumps(spr))
    assert clone._eng is None and len(clone._d) == 0 and isinstance(clone.X, np.ndarray) and clone.X.dtype == X.dtype
    np.testing.assert_array_equal(clone.Ur, spr.Ur)
    assert clone.Ur.dtype == spr.Ur.dtype
    a1, _ = clone.predict(y)
    np.testing.assert_array_equal(a1, a0)
    np.testing.assert_array_equal(clone.reconstruct(a1), x0)
    clone.fit(select_modes='number', n_modes=12)
    clone.optimal_placement()
    np.testing.assert_array_equal(clone.sensors_, spr.sensors_)
    np.testing.assert_array_equal(clone.Sigma_r, spr.Sigma_r)
    assert clone.Ur.dtype == spr.Ur.dtype


def test_public_gem_method_on_the_device(eng):
    """SPR.gem(Ur, ...) (reference :586-698) with the fitted basis and with a foreign one, on the HIP engine"""
    from openmeasure_amd.sparse_sensing import SPR
    rng = np.random.default_rng(4)
    n_points, F, r = 2500, 2, 8
    X = synth_host(n_points, F, 20, 16, 0.8, 1e-3, 9)
    xyz = rng.random((n_points, 3))
    spr = SPR(X, F, xyz, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    C = spr.optimal_placement(calc_type='gem', n_sensors=6)
    np.testing.assert_array_equal(spr.gem(spr.Ur, 6, None, 0.0, False), np.argmax(np.asarray(C), axis=1))
    U = rng.standard_normal((n_points * F, 5))
    want, _ = orc.gem_pivots(U, 4, xyz, F, None, 0.02)
    Ur0 = spr.Ur.copy()
    np.testing.assert_array_equal(spr.gem(U, 4, None, 0.02, False), want)
    assert spr.r == r
    np.testing.assert_array_equal(spr.Ur, Ur0)
    np.testing.assert_array_equal(spr.reconstruct(spr.Ar[0, :])[:, 0], spr.reconstruct(spr.Ar[:1])[:, 0])


def test_gem_sign_dependence_on_the_device(eng, capsys):
    """the GEM sensors depend on the signs of the basis columns (reference :622, :638 take row variances over
    the r entries) -- flip one column and BOTH the HIP path and the oracle's literal formulas move to the same new sensors;
    the dependence is the algorithm's, not the kernels'.  Also gem(verbose=True): the reference's table."""
    from openmeasure_amd.sparse_sensing import SPR
    rng = np.random.default_rng(5)
    n_points, F, r = 4000, 2, 8
    n = n_points * F
    U = rng.standard_normal((n, r)) * (0.5 + rng.random((n, 1))) / np.sqrt(n)
    xyz = rng.random((n_points, 3))
    U2 = U.copy()
    U2[:, 2] *= -1
    want, lead = orc.gem_pivots(U, 6, xyz, F, None, 0.0)
    want2, lead2 = orc.gem_pivots(U2, 6, xyz, F, None, 0.0)
    assert min(lead.min(), lead2.min()) > 1e-3 and not np.array_equal(want, want2)
    spr = SPR(np.zeros((n, r + 1)), F, xyz, engine=eng)
    np.testing.assert_array_equal(spr.gem(U, 6, None, 0.0, False), want)
    np.testing.assert_array_equal(spr.gem(U2, 6, None, 0.0, True), want2)
    out = capsys.readouterr().out.splitlines()
    rows = [ln.split() for ln in out if ln.strip() and ln.split()[0].isdigit()]
    assert len(rows) == 6 and 'sigma^2 y|a' in out[1]
    # ... and the same through fit(basis=...) + optimal_placement, the route a user of the reference's basis takes
    spr.fit(basis=(U2, np.eye(r + 1, r)))
    spr.optimal_placement(calc_type='gem', n_sensors=6)
    np.testing.assert_array_equal(spr.sensors_, want2)


def test_masked_placement_vs_oracle(eng):
    from openmeasure_amd.sparse_sensing import SPR
    X = synth_host(3000, 3, 24, 24, 0.75, 1e-3, 77)
    st = orc.fit(X, 3, 'number', 8)
    mask = np.random.default_rng(3).random(X.shape[0]) < 0.4
    piv, Ur_m = orc.qr_pivots(st['Ur'], mask)
    spr = SPR(X, 3, None, engine=eng)
    spr.fit(select_modes='number', n_modes=8)
    spr.optimal_placement(mask=mask)
    np.testing.assert_array_equal(spr.sensors_, piv)
    assert mask[spr.sensors_].all() and not spr.Ur[~mask].any()


def test_duplicate_rows_tie_goes_to_lowest_index(eng):
    """exact ties (duplicated cells) must resolve like LAPACK's idamax: first index wins"""
    rng = np.random.default_rng(5)
    U = rng.standard_normal((500, 6))
    U[400] = U[17] = U[3] * 0 + 5.0 * rng.standard_normal(6)   # two identical dominant rows
    Ud = eng.to_device(U)
    from openmeasure_amd.sparse_sensing import pivot_loop
    st = eng.qr_begin(Ud, 0, 6)
    pivot_loop(eng, st, 6)
    piv = eng.to_host(st['piv'])
    ref, _ = orc.qr_pivots(U)
    assert piv[0] == 17 and 400 not in piv[:1]
    np.testing.assert_array_equal(piv, ref)


@pytest.mark.parametrize('n,r,seed', [(200000, 32, 1), (50000, 64, 2), (3000, 14, 3), (700000, 8, 4),
                                      (20000, 200, 5), (9000, 300, 6), (6000, 131, 7), (5000, 512, 8), (3000, 1024, 9)])
def test_pivots_candidate_set_vs_oracle(eng, n, r, seed):
    """orthonormal random basis: row norms are nearly uniform, so the candidate bound is tight and
    certification fails often -- the batches must still reproduce dgeqp3's order exactly"""
    from openmeasure_amd.sparse_sensing import pivot_loop
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.standard_normal((n, r)))
    ref, _ = orc.qr_pivots(U)
    Ud = eng.to_device(U)
    st = eng.qr_begin(Ud, 0, r)
    sweeps = pivot_loop(eng, st, r)
    np.testing.assert_array_equal(eng.to_host(st['piv']), ref)
    assert 1 <= sweeps <= r


@pytest.mark.parametrize('n,r,seed', [(200000, 32, 1), (50000, 64, 2), (700000, 16, 4), (30000, 128, 6), (120000, 48, 7)])
def test_pivots_pooled_vs_oracle(eng, n, r, seed):
    """the same near-uniform bases through the epoch-sweep driver (pools=True): thresholds that rarely pay, many failed
    certifications, every refresh an epoch sweep -- dgeqp3's order exactly"""
    from openmeasure_amd.sparse_sensing import pivot_loop
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.standard_normal((n, r)))
    ref, _ = orc.qr_pivots(U)
    Ud = eng.to_device(U)
    st = eng.qr_begin(Ud, 0, r)
    stats = {}
    sweeps = pivot_loop(eng, st, r, pools=True, stats=stats)
    np.testing.assert_array_equal(eng.to_host(st['piv']), ref)
    assert 1 <= sweeps <= r and ('pool_sweeps' in stats) == (r > eng.qr_batch)   # one batch: nothing to refresh


@pytest.mark.parametrize('seed', range(18))
def test_placement_drivers_random_bases(eng, seed):
    """Random bases through both placement drivers (refresh per batch; epoch sweeps with pools and one launch per step):
    r from 16 to 128, 5k to 400k rows, iid / heavy-tailed / spatially clustered row norms, f64 and f32 storage, a shard
    offset -- the order of dgeqp3 on the stored values, every time."""
    import torch
    from openmeasure_amd.sparse_sensing import pivot_loop
    rng = np.random.default_rng(9000 + seed)
    r = int([16, 32, 48, 64, 96, 128][seed % 6])
    n = int(rng.integers(5_000, 400_000 if r <= 64 else 120_000))
    kind = ['iid', 'heavy', 'clustered'][(seed // 2) % 3]
    U = rng.standard_normal((n, r))
    if kind == 'heavy':
        U *= np.exp(rng.standard_normal((n, 1)))
    elif kind == 'clustered':                      # the large rows sit together, as in a field with a localised feature
        U *= (1.0 + 8.0 * np.exp(-((np.arange(n) - 0.37 * n) / (0.01 * n)) ** 2))[:, None]
    U /= np.sqrt(n)
    f32 = seed % 5 == 3
    if f32:
        U = U.astype(np.float32).astype(np.float64)
    ref, _ = orc.qr_pivots(U)
    Ud = eng.to_device(U.astype(np.float32), dtype=torch.float32) if f32 else eng.to_device(U)
    row0 = int(rng.integers(0, 1000)) if seed % 2 else 0
    for pools in (False, True):
        st = eng.qr_begin(Ud, row0, r)
        stats = {}
        sweeps = pivot_loop(eng, st, r, pools=pools, stats=stats)
        np.testing.assert_array_equal(eng.to_host(st['piv']) - row0, ref, err_msg=f'{kind} n={n} r={r} f32={f32} pools={pools}')
        assert 1 <= sweeps <= 1 + (r - 1) // eng.qr_batch + r // 4


def test_pool_build_is_the_sorted_set_of_rows_above_the_threshold(eng):
    import torch
    rng = np.random.default_rng(3)
    for n, theta in ((1, 0.5), (255, 0.9), (70001, 0.97), (1_000_003, 0.5), (1_000_003, 0.999), (300000, 2.0)):
        v = rng.random(n)
        v[rng.integers(0, n, size=max(1, n // 50))] = -1.0             # rows already picked
        st = dict(nrm_e=eng.to_device(v), n=n)
        got = eng.qr_pool_build(st, theta)
        want = np.flatnonzero(v > theta)
        cap = st['pool'].shape[0]
        if len(want) > cap:
            assert got == -1
            continue
        assert got == len(want)
        np.testing.assert_array_equal(eng.to_host(st['pool'][:got].to(torch.int64)), want)


@pytest.mark.parametrize('n,r,f32', [(70000, 64, False), (50001, 128, True), (33333, 16, False), (40000, 48, False)])
def test_epoch_sweep_vs_numpy(eng, n, r, f32):
    """spr_qr_epoch_sweep_*: residual = epoch norm - sum over the epoch's directions of (u . q)^2, for the pool's rows only
    (others untouched, tau floored) or for every row (epoch norms rewritten); picks leave the race; 1 .. 64 directions."""
    import torch
    rng = np.random.default_rng(r)
    U = rng.standard_normal((n, r)) / np.sqrt(r)
    if f32:
        U = U.astype(np.float32).astype(np.float64)
    Ud = eng.to_device(U.astype(np.float32), dtype=torch.float32) if f32 else eng.to_device(U)
    Qh, _ = np.linalg.qr(rng.standard_normal((r, r)))
    dmax = int(eng.lib.spr_qr_epoch_max_directions(r))
    for j_e, j in ((0, 5), (3, min(r, 3 + min(16, dmax))), (0, min(dmax, r)), (7, 7 + min(dmax, r - 7))):
        st = eng.qr_begin(Ud, 11, r)
        eng.qr_epoch_begin(st)
        st['Q'].copy_(eng.to_device(Qh.T.copy()))                          # directions = rows of Q
        picks = rng.choice(n, size=j, replace=False).astype(np.int64) + 11
        st['piv'][:j].copy_(eng.to_device(picks, dtype=torch.int64))
        nrm_e = eng.to_host(st['nrm_e']).copy()
        np.testing.assert_allclose(nrm_e, (U ** 2).sum(axis=1), rtol=1e-13)
        theta = float(np.quantile(nrm_e, 0.9))
        pn = eng.qr_pool_build(st, theta)
        assert pn == int((nrm_e > theta).sum())
        before = eng.to_host(st['nrm']).copy()
        d2 = ((U @ Qh[:, j_e:j]) ** 2).sum(axis=1)
        want = np.maximum(nrm_e - d2, 0.0)
        # pool sweep: pool rows recomputed, the others untouched (apart from the picks), tau >= theta
        eng.qr_epoch_sweep(st, j_e, j, 0, pool=True, tau_floor=theta)
        got = eng.to_host(st['nrm'])
        inpool = nrm_e > theta
        loc = picks - 11
        exp = np.where(inpool, want, before)
        exp[loc] = -1.0
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-13 * nrm_e.max())
        assert float(eng.to_host(st['tau'])[0]) >= theta
        np.testing.assert_array_equal(eng.to_host(st['nrm_e'])[~np.isin(np.arange(n), loc)], nrm_e[~np.isin(np.arange(n), loc)])
        # full sweep: every row, epoch norms rewritten
        eng.qr_epoch_sweep(st, j_e, j, j)
        exp = want.copy()
        exp[loc] = -1.0
        np.testing.assert_allclose(eng.to_host(st['nrm']), exp, rtol=0, atol=1e-13 * nrm_e.max())
        np.testing.assert_array_equal(eng.to_host(st['nrm_e']), eng.to_host(st['nrm']))
        best = eng.to_host(st['rec'])
        assert abs(best[0] - np.max(exp)) <= 1e-13 * nrm_e.max()
        if np.max(exp) > 1e-6 * nrm_e.max():                                  # all r directions applied: rounding noise only
            assert int(best[1]) - 11 == int(np.argmax(exp))


@pytest.mark.parametrize('n_points,F,m,r,f32_basis', [(60000, 4, 256, 64, False), (50000, 4, 64, 32, False),
                                                    (20000, 16, 512, 128, True), (30000, 3, 128, 48, False)])
def test_placement_pools_same_sensors(eng, n_points, F, m, r, f32_basis):
    """SPR.placement_pools: the epoch-sweep driver picks the sensors of the refresh-per-batch driver (and of the oracle),
    with fewer passes over the basis."""
    import torch
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
    rho = 10 ** (-3 / (r - 1))
    X = synth_host(n_points, F, m, min(m, 2 * r), rho, 1e-3, 9 + m)
    Xin = DeviceMatrix(eng.to_device(X.astype(np.float32), dtype=torch.float32), basis='f32') if f32_basis else X
    spr = SPR(Xin, F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    spr.placement_pools = False
    spr.optimal_placement()
    plain, plain_sweeps, gaps = spr.sensors_.copy(), spr.pivot_sweeps_, spr.pivot_gap_.copy()
    assert spr.pivot_pool_sweeps_ == 0
    spr.placement_pools = True
    spr.optimal_placement()
    np.testing.assert_array_equal(spr.sensors_, plain)
    # the gap of a step is measured against the runner-up AMONG THE CANDIDATES: the same wherever that row is a candidate of
    # both drivers (most steps), never a tie
    assert (spr.pivot_gap_ > 0).all() and np.mean(np.isclose(spr.pivot_gap_, gaps, rtol=1e-6)) > 0.5
    assert spr.pivot_sweeps_ <= plain_sweeps
    if r >= 64:      # enough batches for a pool to pay (two or three: the threshold is reached before the first refresh)
        assert spr.pivot_pool_sweeps_ >= 1 and spr.pivot_sweeps_ < plain_sweeps, (spr.pivot_sweeps_, spr.pivot_pool_sweeps_, plain_sweeps)
    if not f32_basis:
        np.testing.assert_array_equal(plain, orc.qr_pivots(orc.fit(X, F, 'number', r)['Ur'])[0])


def test_weighted_predict_batch_vs_oracle(eng):
    rng = np.random.default_rng(8)
    s, r, F, n_p = 40, 12, 3, 4
    Theta = rng.standard_normal((s, r))
    cnt = rng.standard_normal(s)
    scl_f = 1 + rng.random(F)
    ys = []
    for p in range(n_p):
        y = np.zeros((s, 3))
        y[:, 0] = rng.standard_normal(s)
        y[:, 2] = rng.integers(0, F, s)
        if p % 2:
            y[:, 1] = 0.05 + rng.random(s)
        ys.append(y)
    n_points = 7
    X_cnt = np.zeros((F * n_points, 1))
    X_scl = np.repeat(scl_f, n_points)[:, None]

    class _C:                                            # C.dot(X_cnt) -> cnt
        def dot(self, v):
            return cnt
    Ar_ref, As_ref = orc.predict_ols(ys, Theta, _C(), X_cnt, X_scl, n_points)
    Ar, As, y0, info = eng.solve_ols(eng.to_device(Theta), eng.to_device(cnt), eng.to_device(scl_f),
                                     eng.to_device(np.stack(ys)))
    np.testing.assert_allclose(eng.to_host(Ar), Ar_ref, rtol=0, atol=1e-10 * np.abs(Ar_ref).max())
    np.testing.assert_allclose(eng.to_host(As), As_ref, rtol=0, atol=1e-10 * np.abs(As_ref).max())
    assert not eng.to_host(info)[:, 0].any()


@pytest.mark.parametrize('cond,tol', [(1e2, 1e-12), (1e4, 1e-10), (1e5, 1e-9), (1e6, 1e-7)])
def test_predict_moderately_ill_conditioned_theta(eng, cond, tol):
    """Normal equations alone lose cond^2 eps; with the refinement step of solve.hip the coefficients agree with the
    oracle's SVD pseudo-inverse to about cond eps (weighted and unweighted right-hand sides)."""
    import torch
    rng = np.random.default_rng(11)
    s_, r = 48, 24
    Uq, _ = np.linalg.qr(rng.standard_normal((s_, r)))
    Vq, _ = np.linalg.qr(rng.standard_normal((r, r)))
    Theta = (Uq * np.logspace(0, -np.log10(cond), r)) @ Vq.T
    a_true = rng.standard_normal(r)
    ys = []
    for weighted in (False, True):
        y = np.zeros((s_, 3))
        y[:, 0] = Theta @ a_true + 1e-3 * rng.standard_normal(s_)
        if weighted:
            y[:, 1] = 0.05 * (1.0 + rng.random(s_))
        ys.append(y)
    Y = eng.to_device(np.stack(ys))
    Ar, As, _, info = eng.solve_ols(eng.to_device(Theta), eng.to_device(np.zeros(s_)), eng.to_device(np.ones(1)), Y)
    Ar, As, info = eng.to_host(Ar), eng.to_host(As), eng.to_host(info)
    assert not info[:, 0].any()
    C = np.eye(s_)                                             # identity measurement: cnt = 0, scl = 1, feature 0
    A_ref, S_ref = orc.predict_ols(ys, Theta, C, np.zeros((s_, 1)), np.ones((s_, 1)), s_)
    for k in range(2):
        assert np.linalg.norm(Ar[k] - A_ref[k]) <= tol * np.linalg.norm(A_ref[k]), (k, info[k])
    assert np.linalg.norm(As[1] - S_ref[1]) <= tol * np.linalg.norm(S_ref[1])


@pytest.mark.parametrize('decades', [3, 5, 7, 9, 11, 13])
def test_conditioning_guard(eng, decades):                # sigma_1/sigma_r up to 1e13: exact sensors or LinAlgError
    run_conditioning_guard(eng, decades, synth_host)
    if decades in (5, 7):                                  # f32-stored data: refinement through spr_project_x32_f64out
        run_conditioning_guard(eng, decades, synth_host, f32=True)
    if decades == 7:                                        # wider shapes: two column slices (m = 300), W in two column groups (m = 160)
        run_conditioning_guard(eng, decades, synth_host, shape=(600, 2, 300, 12))
        run_conditioning_guard(eng, decades, synth_host, shape=(900, 3, 160, 24))
        run_conditioning_guard(eng, 5, synth_host, f32=True, shape=(500, 2, 300, 9))
        run_conditioning_guard(eng, decades, synth_host, shape=(400, 2, 600, 12))       # slice-pair Gram + streamed-W refinement


@pytest.mark.parametrize('n_points,F,r,n_sensors,d_min,masked', [(150, 2, 5, 9, 0.0, False), (2000, 3, 16, 24, 0.03, True),
                                                              (5000, 2, 8, 40, 0.0, False), (700, 4, 33, 40, 0.02, False)])
def test_gem_beyond_rank(eng, n_points, F, r, n_sensors, d_min, masked):   # more sensors than r-1: ridge rule vs oracle
    run_gem_beyond_rank(eng, n_points, F, r, n_sensors, d_min, masked, 50 + r)


def test_pinv_fixture(golden_pinv, eng):                  # :873-878 -- underdetermined / rank-deficient / ill-conditioned
    run_pinv_fixture(golden_pinv, eng)


def _identity_problem(Theta, ys):
    """oracle predict_ols for a bare Theta: identity measurement matrix, zero centre, unit scale"""
    s_ = Theta.shape[0]
    return orc.predict_ols(ys, Theta, np.eye(s_), np.zeros((s_, 1)), np.ones((s_, 1)), s_)


@pytest.mark.parametrize('s_,r,rank', [(5, 12, 5), (12, 12, 9), (40, 17, 11), (200, 33, 33), (70, 64, 40), (300, 128, 100),
                                      (130, 128, 128), (3, 128, 3), (1, 4, 1), (9, 1, 1),
                                      (150, 200, 150), (200, 200, 160), (450, 300, 300), (300, 300, 300), (40, 513, 33),
                                      (600, 600, 600)])
def test_pinv_kernel_vs_oracle(eng, s_, r, rank):
    """spr_solve_pinv_f64 against np.linalg.pinv on random systems of prescribed rank (exact low rank through a
    product of thin factors; s < r, s = r, s > r; all four r classes of the kernel), weighted and unweighted."""
    rng = np.random.default_rng(100 + s_ + r)
    Theta = rng.standard_normal((s_, rank)) @ rng.standard_normal((rank, r)) if rank < min(s_, r) else \
        rng.standard_normal((s_, r))
    ys = []
    for weighted in (False, True, True):
        y = np.zeros((s_, 3))
        y[:, 0] = rng.standard_normal(s_)
        if weighted:
            y[:, 1] = 0.05 * (1.0 + rng.random(s_))
        ys.append(y)
    A_ref, S_ref = _identity_problem(Theta, ys)
    Ar, As, y0, info = eng.solve_pinv(eng.to_device(Theta), eng.to_device(np.zeros(s_)), eng.to_device(np.ones(1)),
                                      eng.to_device(np.stack(ys)))
    Ar, As, info = eng.to_host(Ar), eng.to_host(As), eng.to_host(info)
    assert (info[:, 0] >= (1 if min(s_, r) > 1 else 0)).all() and (info[:, 0] <= 30).all(), info   # Jacobi converged
    np.testing.assert_array_equal(info[:, 1], min(rank, s_, r))               # numerical rank = the designed one
    # exact rank deficiency: numpy's own tiny singular values sit at eps * sigma_max, a factor ~5 under its cut;
    # both solutions are the minimum-norm one up to that noise
    for k in range(3):
        assert np.linalg.norm(Ar[k] - A_ref[k]) <= 1e-9 * np.linalg.norm(A_ref[k]), (k, info[k])
    assert np.linalg.norm(As[1] - S_ref[1]) <= 1e-9 * np.linalg.norm(S_ref[1])
    assert not As[0].any()
    np.testing.assert_allclose(eng.to_host(y0)[1, :, 1], ys[1][:, 1])


@pytest.mark.parametrize('cond', [1e7, 1e10, 1e13])
def test_predict_ill_conditioned_goes_through_svd(eng, cond):
    """Beyond cond ~ 3e6 the normal equations are refused by predict() and the QR + Jacobi-SVD kernel answers, as
    np.linalg.pinv does, to about cond * eps."""
    rng = np.random.default_rng(17)
    s_, r = 60, 20
    Uq, _ = np.linalg.qr(rng.standard_normal((s_, r)))
    Vq, _ = np.linalg.qr(rng.standard_normal((r, r)))
    Theta = (Uq * np.logspace(0, -np.log10(cond), r)) @ Vq.T
    y = np.zeros((s_, 3))
    y[:, 0] = Theta @ rng.standard_normal(r)
    A_ref, _ = _identity_problem(Theta, [y])
    Ar, _, _, info = eng.solve_pinv(eng.to_device(Theta), eng.to_device(np.zeros(s_)), eng.to_device(np.ones(1)),
                                    eng.to_device(y[None]))
    info = eng.to_host(info)
    assert info[0, 1] == r
    np.testing.assert_allclose(info[0, 2] / info[0, 3], cond, rtol=1e-3 if cond < 1e12 else 0.2)
    assert np.linalg.norm(eng.to_host(Ar)[0] - A_ref[0]) <= 50 * cond * 2.2e-16 * np.linalg.norm(A_ref[0])


def test_predict_tomography_sized_weighted_solve(eng):
    """s = 4096 sensors x r = 32 modes (many panels of the MFMA normal equations), weighted, batch of 3 -- and the
    same systems through the streaming-QR pinv kernel."""
    rng = np.random.default_rng(23)
    s_, r, F = 4096, 32, 5
    Theta = rng.standard_normal((s_, r)) / np.sqrt(s_)
    cnt = rng.standard_normal(s_)
    scl_f = 1 + rng.random(F)
    ys = []
    for p in range(3):
        y = np.zeros((s_, 3))
        y[:, 0] = rng.standard_normal(s_)
        y[:, 2] = rng.integers(0, F, s_)
        if p:
            y[:, 1] = 0.05 + rng.random(s_)
        ys.append(y)
    n_points = 3
    X_scl = np.repeat(scl_f, n_points)[:, None]

    class _C:
        def dot(self, v):
            return cnt
    A_ref, S_ref = orc.predict_ols(ys, Theta, _C(), np.zeros((F * n_points, 1)), X_scl, n_points)
    args = (eng.to_device(Theta), eng.to_device(cnt), eng.to_device(scl_f), eng.to_device(np.stack(ys)))
    Ar, As, _, info = eng.solve_ols(*args)
    assert not eng.to_host(info)[:, 0].any()
    np.testing.assert_allclose(eng.to_host(Ar), A_ref, rtol=0, atol=1e-10 * np.abs(A_ref).max())
    np.testing.assert_allclose(eng.to_host(As), S_ref, rtol=0, atol=1e-10 * np.abs(S_ref).max())
    Ar2, As2, _, info2 = eng.solve_pinv(*args)
    assert (eng.to_host(info2)[:, 1] == r).all()
    np.testing.assert_allclose(eng.to_host(Ar2), A_ref, rtol=0, atol=1e-10 * np.abs(A_ref).max())
    np.testing.assert_allclose(eng.to_host(As2), S_ref, rtol=0, atol=1e-10 * np.abs(S_ref).max())


def test_predict_feature_id_out_of_range_raises_like_the_reference(eng):
    """:576 indexes X_scl with y[:,2]*n_points: an id past the last feature is an IndexError, a negative one wraps."""
    X = synth_host(300, 3, 12, 8, 0.7, 1e-3, 5)
    from openmeasure_amd.sparse_sensing import SPR
    spr = SPR(X, 3, None, engine=eng)
    spr.fit(select_modes='number', n_modes=4)
    C = spr.optimal_placement()
    spr.train(C)
    y = np.zeros((4, 3)); y[:, 0] = X[spr.sensors_, 0]; y[:, 2] = spr.sensors_ // 300
    a_ok, _ = spr.predict(y)
    ybad = y.copy(); ybad[1, 2] = 3
    with pytest.raises(IndexError):
        spr.predict(ybad)
    yneg = y.copy(); yneg[:, 2] -= 3                                      # -3..-1 wrap onto features 0..2
    a_neg, _ = spr.predict(yneg)
    np.testing.assert_array_equal(a_neg, a_ok)
    # a Theta handed over directly with another sensor count no longer matches C (ADVICE): the reference fails in
    # scale_vector's broadcast, never reads out of bounds
    spr.train(np.vstack([spr.Theta, spr.Theta]), is_Theta=True)
    with pytest.raises(ValueError):
        spr.predict(np.vstack([y, y]))


def test_reconstruct_many_vectors(eng):
    """n_p = 40 coefficient vectors: three passes of 16 + 16 + 8 through reconstruct.hip, f64 and f32-stored basis."""
    import torch
    rng = np.random.default_rng(31)
    n_points, F, r, n_p = 4099, 3, 24, 40
    n = n_points * F
    U = rng.standard_normal((n, r))
    mu = rng.standard_normal(n)
    scl_f = 1 + rng.random(F)
    A = rng.standard_normal((n_p, r))
    ref = orc.reconstruct(A, U, mu[:, None], np.repeat(scl_f, n_points)[:, None])
    out = eng.reconstruct(eng.to_device(U), 0, n_points, F, eng.to_device(mu), eng.to_device(scl_f), eng.to_device(A))
    got = eng.to_host(out).T
    assert got.shape == (n, n_p)
    assert rel_fro(got, ref) <= 1e-14
    U32 = U.astype(np.float32)
    ref32 = orc.reconstruct(A, U32.astype(np.float64), mu[:, None], np.repeat(scl_f, n_points)[:, None])
    out32 = eng.reconstruct(eng.to_device(U32, dtype=torch.float32), 0, n_points, F, eng.to_device(mu),
                            eng.to_device(scl_f), eng.to_device(A))
    assert rel_fro(eng.to_host(out32).T, ref32) <= 1e-14


def test_csr_measurement_matrix_with_many_nonzeros(eng):
    """tomography-like C: 600 rays x 50 000 cells with 2e5 non-zeros (CSR measure kernel, Theta and cnt)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(41)
    n, r, s_ = 50_000, 32, 600
    U = rng.standard_normal((n, r))
    mu = rng.standard_normal(n)
    C = sp.random(s_, n, density=2e5 / (s_ * n), random_state=9, format='csr')
    assert C.nnz >= 1e5
    t = eng.torch
    Th, cnt = eng.measure_csr(eng.to_device(C.indptr, dtype=t.int64), eng.to_device(C.indices, dtype=t.int64),
                              eng.to_device(C.data), eng.to_device(U), 0, eng.to_device(mu))
    np.testing.assert_allclose(eng.to_host(Th), C @ U, atol=1e-11)
    np.testing.assert_allclose(eng.to_host(cnt), C @ mu, atol=1e-11)


def test_general_csr_measurement_matrix(eng):
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    n, r, s = 5000, 24, 37
    U = rng.standard_normal((n, r))
    mu = rng.standard_normal(n)
    C = sp.random(s, n, density=0.01, random_state=4, format='csr')
    t = eng.torch
    Th, cnt = eng.measure_csr(eng.to_device(C.indptr, dtype=t.int64), eng.to_device(C.indices, dtype=t.int64),
                              eng.to_device(C.data), eng.to_device(U), 0, eng.to_device(mu))
    np.testing.assert_allclose(eng.to_host(Th), C @ U, atol=1e-12)
    np.testing.assert_allclose(eng.to_host(cnt), C @ mu, atol=1e-12)


@pytest.mark.parametrize('n_points,F,m,r,scale_type', [(500, 3, 12, 4, 'std'), (800, 9, 41, 14, 'pareto'), (600, 4, 64, 32, 'std'),
                                                        (300, 2, 7, 7, 'l2-norm'), (400, 70, 16, 5, 'vast')])
def test_device_spectrum_vs_lapack(eng, n_points, F, m, r, scale_type):
    """spr_spectrum_f64 (Jacobi, m <= 64) against numpy.linalg.eigh on the same scaled Gram matrix, and the
    feature scales against the oracle's scale_data"""
    X = synth_host(n_points, F, m, min(m, 2 * r), 0.8, 1e-3, 4000 + m) * 0.05 + 5.0
    X_cnt, X_scl, X0 = orc.scale_data(X, F, scale_type)
    rowmean, fstats, gram = eng.stats_gram(eng.to_device(X), 0, n_points, F)
    sp = eng.spectrum(gram, fstats[None], scale_type, r)
    np.testing.assert_allclose(eng.to_host(sp['scale']), X_scl[::n_points, 0], rtol=1e-11)
    S_ref = np.linalg.svd(X0, compute_uv=False)
    S = eng.to_host(sp['S'])
    rw = int(np.sum(S_ref > 1e-6 * S_ref[0]))
    np.testing.assert_allclose(S[:rw], S_ref[:rw], rtol=1e-8)
    V = eng.to_host(sp['V'])
    assert np.abs(V.T @ V - np.eye(m)).max() < 1e-12
    G = X0.T @ X0
    assert np.abs(V.T @ G @ V - np.diag(eng.to_host(sp['lam']))).max() <= 1e-10 * S_ref[0] ** 2
    rr = min(r, rw)
    W = eng.to_host(sp['W'])
    np.testing.assert_allclose(W[:, :rr] * S[:rr], V[:, :rr], atol=1e-12)
    assert eng.to_host(sp['info'])[0] <= 15


# ---- row shards: the kernels see rows [row0, row0+n_loc) of a global feature-major matrix ----
@pytest.mark.parametrize('n_points,F,m,r,world', [(1000, 3, 24, 8, 2), (777, 4, 64, 16, 3), (500, 9, 40, 14, 4), (64, 2, 256, 32, 2)])
def test_kernels_on_row_shards(eng, n_points, F, m, r, world):
    """Every shard kernel with row0 != 0 and features that straddle the shard boundaries; the
    per-shard results must add up / concatenate to the single-shard ones and to the oracle."""
    from tests.numpy_engine import NumpyEngine
    X = synth_host(n_points, F, m, min(m, 2 * r), 0.85, 1e-3, 900 + m)
    n = n_points * F
    if n % world:
        pytest.skip('rows do not divide')
    n_loc = n // world
    X_cnt, X_scl, X0 = orc.scale_data_std(X, F)
    ref = NumpyEngine()
    W = np.random.default_rng(2).standard_normal((m, r))
    inv = 1.0 / X_scl[::n_points, 0]
    gram_sum = np.zeros((F, m, m)); cnt = np.zeros(F); Us = []; recs = []; means = []
    for k in range(world):
        row0 = k * n_loc
        Xs = eng.to_device(X[row0:row0 + n_loc])
        rowmean, fstats, gram = eng.stats_gram(Xs, row0, n_points, F)
        rm_ref, fs_ref, g_ref = ref.stats_gram(ref.to_device(X[row0:row0 + n_loc]), row0, n_points, F)
        np.testing.assert_allclose(eng.to_host(rowmean), rm_ref.numpy(), rtol=1e-13, atol=1e-13)
        fs = eng.to_host(fstats)
        np.testing.assert_array_equal(fs[:, 0], fs_ref.numpy()[:, 0])
        np.testing.assert_allclose(fs[:, 1:], fs_ref.numpy()[:, 1:], rtol=1e-9, atol=1e-9 * np.abs(fs_ref.numpy()).max())
        g = eng.to_host(gram)
        assert np.abs(g - g_ref.numpy()).max() <= 1e-12 * max(np.abs(g_ref.numpy()).max(), 1e-300)
        gram_sum += g; cnt += fs[:, 0]
        U = eng.project(Xs, row0, n_points, F, eng.to_device(inv), eng.to_device(W), rowmean=rowmean)
        Us.append(eng.to_host(U)); means.append(eng.to_host(rowmean))
        A = np.random.default_rng(3).standard_normal((2, r))
        xr = eng.to_host(eng.reconstruct(U, row0, n_points, F, rowmean, eng.to_device(X_scl[::n_points, 0]), eng.to_device(A))).T
        assert rel_fro(xr, orc.reconstruct(A, Us[-1], X_cnt[row0:row0 + n_loc], X_scl[row0:row0 + n_loc])) <= 1e-13
        st = eng.qr_begin(U, row0, r)
        recs.append(eng.to_host(st['rec']))
    np.testing.assert_array_equal(cnt, n_points)
    Ufull = np.vstack(Us)
    assert np.abs(Ufull - X0 @ W).max() <= 1e-11 * np.abs(X0 @ W).max()
    # the best record over the shards is the globally largest row, with its GLOBAL index
    nrm = (Ufull ** 2).sum(axis=1)
    best = max(recs, key=lambda c: (c[0], -c[1]))
    assert int(best[1]) == int(np.argmax(nrm)) and abs(best[0] - nrm.max()) <= 1e-12 * nrm.max()


@pytest.mark.parametrize('n_points,F,m,r', [(1, 1, 2, 1), (3, 2, 2, 2), (17, 1, 1, 1), (5, 3, 300, 2), (5, 3, 600, 2)])
def test_tiny_and_out_of_range_shapes(eng, n_points, F, m, r):
    """shapes far below one panel / one workgroup, narrow and wide (slice pairs for m > 512: nothing is refused)"""
    rng = np.random.default_rng(n_points + m)
    X = rng.standard_normal((n_points * F, m)) + 3.0
    Xd = eng.to_device(X)
    rowmean, fstats, gram = eng.stats_gram(Xd, 0, n_points, F)
    np.testing.assert_allclose(eng.to_host(rowmean), X.mean(axis=1), rtol=1e-14)
    c = X - X.mean(axis=1, keepdims=True)
    for f in range(F):
        blk = c[f * n_points:(f + 1) * n_points]
        np.testing.assert_allclose(eng.to_host(gram)[f], blk.T @ blk, atol=1e-13 * max(1.0, np.abs(blk).max() ** 2))
    W = rng.standard_normal((m, r))
    U = eng.to_host(eng.project(Xd, 0, n_points, F, eng.to_device(np.ones(F)), eng.to_device(W), rowmean=rowmean))
    np.testing.assert_allclose(U, c @ W, atol=1e-12)


# ---- size-independent properties at a larger size (too big for the oracle's full path in seconds) ----
def test_properties_at_scale(eng):
    """1M cells x 4 features x 64 snapshots generated on the device (BASELINE config 2 shape):
    U_r^T U_r = I, Gram trace identity, reconstruct(Ar[j]) == X[:, j] restricted to the span,
    and run-to-run determinism of the pivots."""
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
    from openmeasure_amd.synth import make_R
    n_points, F, m, r = 1_000_000, 4, 64, 32
    R = eng.to_device(make_R(m, r, seed=1234))
    Xd = eng.synth(n_points * F, m, 0, n_points, R, 1e-3, 1234)
    spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    t = eng.torch
    Ur = spr._d['Ur']
    I = eng.to_host(Ur.T @ Ur)                             # torch only as the checker here
    assert np.abs(I - np.eye(r)).max() < 1e-9
    assert spr.S_[0] / spr.S_[r - 1] < 1e4                 # designed spectrum
    # total variance: sum_f trace(G_f)/var_f == n*m - (row-mean part)  <=> sum(S^2) == ||X0||_F^2
    x0 = (Xd - spr._d['rowmean'][:, None]) * spr._d['inv_scale'].repeat_interleave(n_points)[:, None]
    assert abs(float((x0 * x0).sum()) - float(np.sum(spr.S_ ** 2))) <= 1e-10 * float(np.sum(spr.S_ ** 2))
    # projection round trip: Ur Ar^T is the best rank-r approximation of X0 -> residual energy = tail of S
    rec = spr.reconstruct(spr.Ar[:3], to_host=False)       # (3, n)
    tail = np.sqrt(np.sum(spr.S_[r:] ** 2))
    for j in range(3):
        err = float(t.linalg.norm((rec[j] - Xd[:, j]) / spr._d['scale'].repeat_interleave(n_points)))
        assert err <= 1.05 * tail
    spr.optimal_placement()
    p1 = spr.sensors_.copy()
    spr.optimal_placement()
    np.testing.assert_array_equal(p1, spr.sensors_)
    assert len(set(p1.tolist())) == r and spr.pivot_gap_.min() > 1e-9


def _full_vs_oracle(eng, Xd, F, s):
    """The HIP path on the device matrix Xd against the oracle on the same values downloaded: ordered sensors equal to dgeqp3's
    pivots of the oracle's basis, retained singular values to 1e-8, the reconstructed field within the north_star's 1e-6
    relative Frobenius (same coefficient vector, signs aligned)."""
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
    X = eng.to_host(Xd)
    xr_cpu, st = orc.fit_reconstruct_timed(X, F, s)
    piv_cpu, _ = orc.qr_pivots(st['Ur'])
    spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=s)
    assert spr.r == s == st['r']
    np.testing.assert_allclose(spr.Sigma_r, st['Sigma_r'], rtol=1e-8)
    spr.optimal_placement()
    np.testing.assert_array_equal(spr.sensors_, piv_cpu)                       # exact and ordered
    sg = np.sign(np.sum(spr.Ur * st['Ur'], axis=0))                             # same coefficients need the same column signs
    xr = spr.reconstruct(st['Ar'][0] * sg)
    err = rel_fro(xr[:, 0], xr_cpu.reshape(-1))
    assert err <= REL_FRO, err
    return dict(field_rel_fro=err, sigma_rel=float(np.max(np.abs(spr.Sigma_r - st['Sigma_r']) / st['Sigma_r'])),
                min_pivot_gap=float(spr.pivot_gap_.min()))


def test_config2_full_size_vs_oracle(eng):
    """BASELINE config 2 at FULL size against the oracle (until round 6 this comparison only existed in
    builder-run bench lines): 1M cells x 4 features x 64 snapshots generated on the device by the bench's generator and seed,
    32 modes / sensors; the oracle's SVD of the 4M x 64 matrix takes some 15 s of host LAPACK."""
    from openmeasure_amd.synth import make_R
    n_points, F, m, s = 1_000_000, 4, 64, 32
    R = eng.to_device(make_R(m, s, seed=1234))
    Xd = eng.synth(n_points * F, m, 0, n_points, R, 1e-3, 1234)
    got = _full_vs_oracle(eng, Xd, F, s)
    assert got['min_pivot_gap'] > 1e-9
    print('config 2, full size, HIP vs oracle:', got)


def test_config3_sample_vs_oracle(eng):
    """BASELINE config 3's oracle sample -- the first 100 000 cells of each of the 9 features of the 10M-cell matrix, the very
    rows bench.py's cpu_baseline / parity leg cuts out of the resident shard -- generated here row block by row block with the
    same counter-based generator (its values depend on the GLOBAL row only), 256 snapshots, 64 modes / sensors."""
    import torch
    from openmeasure_amd.synth import make_R
    n_points, cc, F, m, s = 10_000_000, 100_000, 9, 256, 64
    R = eng.to_device(make_R(m, s, seed=1234))
    Xd = torch.cat([eng.synth(cc, m, f * n_points, n_points, R, 1e-3, 1234) for f in range(F)])
    # the generator is what bench.py holds: a block generated on its own equals the same rows cut out of a larger one
    whole = eng.synth(3 * cc, m, 2 * n_points - cc, n_points, R, 1e-3, 1234)      # rows straddling the boundary of features 1 | 2
    assert torch.equal(whole[cc:2 * cc], Xd[2 * cc:3 * cc])
    del whole
    got = _full_vs_oracle(eng, Xd, F, s)
    print('config 3, 100 000-cell sample, HIP vs oracle:', got)


def test_properties_at_config3_scale(eng):
    """BASELINE config 3 at FULL size (10M cells x 9 features x 256 snapshots = 184 GB, 64 modes), generated on the
    device: orthonormal basis, energy identity, best-rank-r residual of a reconstructed training column, sensors
    distinct / reproducible and equal to the oracle's dgeqp3 pivots on the rows both can hold -- all checked in
    10M-row slices (torch only as the checker)."""
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
    from openmeasure_amd.synth import make_R
    t = eng.torch
    t.cuda.empty_cache()
    free, _ = t.cuda.mem_get_info()
    if free < 250e9:
        pytest.skip(f'needs 250 GB of free HBM, {free / 1e9:.0f} GB available')
    n_points, F, m, r = 10_000_000, 9, 256, 64
    n = n_points * F
    R = eng.to_device(make_R(m, r, seed=1234))
    Xd = eng.synth(n, m, 0, n_points, R, 1e-3, 1234)
    spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    Ur, mu, inv = spr._d['Ur'], spr._d['rowmean'], spr._d['inv_scale']
    step = 5_000_000
    gram_u = t.zeros((r, r), dtype=t.float64, device=Ur.device)
    energy = 0.0
    for i0 in range(0, n, step):
        u = Ur[i0:i0 + step]
        gram_u += u.T @ u
        f = i0 // n_points
        assert (i0 + step - 1) // n_points == f                         # slices never straddle a feature here
        x0 = (Xd[i0:i0 + step] - mu[i0:i0 + step, None]) * inv[f]
        energy += float((x0 * x0).sum())
        del x0
    assert np.abs(eng.to_host(gram_u) - np.eye(r)).max() < 1e-9
    assert abs(energy - float(np.sum(spr.S_ ** 2))) <= 1e-10 * energy
    assert spr.S_[0] / spr.S_[r - 1] < 1e4
    rec = spr.reconstruct(spr.Ar[:1], to_host=False)                    # (1, n): training column 0
    tail = np.sqrt(np.sum(spr.S_[r:] ** 2))
    err2 = 0.0
    for i0 in range(0, n, step):
        f = i0 // n_points
        d = (rec[0, i0:i0 + step] - Xd[i0:i0 + step, 0]) * inv[f]
        err2 += float((d * d).sum())
    assert np.sqrt(err2) <= 1.05 * tail
    spr.optimal_placement()
    p1 = spr.sensors_.copy()
    spr.optimal_placement()
    np.testing.assert_array_equal(p1, spr.sensors_)
    assert len(set(p1.tolist())) == r and spr.pivot_gap_.min() > 1e-9
    # the oracle can pivot a slab of the basis: the device picks restricted to that slab must agree with dgeqp3 on it
    slab = slice(0, 2_000_000)
    sub = SPR(np.zeros((2_000_000, r + 2)), 1, None, engine=eng)
    Uh = eng.to_host(Ur[slab])
    sub.fit(basis=(Uh, np.eye(r + 2, r)))
    sub.optimal_placement()
    want, _ = orc.qr_pivots(Uh)
    np.testing.assert_array_equal(sub.sensors_, want)
    del Xd, spr, sub, rec
    t.cuda.empty_cache()


def _slab_pivot_check(eng, Ur_d, r, rows=1_500_000):
    """the oracle (dgeqp3) can pivot a slab of the basis: the device picks restricted to that slab must agree with it"""
    from openmeasure_amd.sparse_sensing import SPR
    Uh = eng.to_host(Ur_d[:rows]).astype(np.float64)
    sub = SPR(np.zeros((rows, r + 2)), 1, None, engine=eng)
    sub.fit(basis=(Uh, np.eye(r + 2, r)))
    sub.optimal_placement()
    want, _ = orc.qr_pivots(Uh)
    np.testing.assert_array_equal(sub.sensors_, want)


def test_properties_at_config4_rank_block(eng):
    """BASELINE config 4 as ONE rank of eight sees it, at full size: rank 3's block of the 10M-cell x 9 x 256 matrix --
    global rows [33.75M, 45M), the end of feature 3 and the start of feature 4 -- through RowShard(partial=True) (what
    bench.py --share-of 8 --share-rank 3 runs): finite spectrum, orthonormal basis ON THE BLOCK'S OWN ROWS (the fit of
    a partial group is the fit of its rows), energy identity per feature segment, best-rank-r residual, reproducible
    sensors and the slab-vs-dgeqp3 pivot check."""
    import bench
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix, RowShard
    from openmeasure_amd.synth import make_R
    t = eng.torch
    t.cuda.empty_cache()
    wl = bench.WORKLOADS['c4']
    F, m, r = wl['features'], wl['m'], wl['s']
    plan = bench.shard_plan(wl, 8, 3)
    n_points, n_glob, n_loc, row0 = plan['n_points'], plan['n_glob'], plan['n_loc'], plan['row0']
    assert (row0, n_loc) == (33_750_000, 11_250_000) and row0 // n_points != (row0 + n_loc - 1) // n_points
    R = eng.to_device(make_R(m, r, seed=1234))
    Xd = eng.synth(n_loc, m, row0, n_points, R, 1e-3, 1234)
    spr = SPR(DeviceMatrix(Xd), F, None, shard=RowShard(row0, n_glob, partial=True), engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    assert np.isfinite(spr.S_).all() and spr.S_[0] / spr.S_[r - 1] < 1e4
    feats = sorted(set([row0 // n_points, (row0 + n_loc - 1) // n_points]))
    absent = [f for f in range(F) if f not in feats]
    np.testing.assert_array_equal(spr._scl_f[absent], 1.0)              # features without rows here take no part
    Ur, mu, inv = spr._d['Ur'], spr._d['rowmean'], spr._d['inv_scale']
    I = eng.to_host(Ur.T @ Ur)
    assert np.abs(I - np.eye(r)).max() < 1e-9
    cut = (feats[0] + 1) * n_points - row0                               # local row where feature 4 starts
    energy = 0.0
    for (a, b, f) in ((0, cut, feats[0]), (cut, n_loc, feats[1])):
        x0 = (Xd[a:b] - mu[a:b, None]) * inv[f]
        energy += float((x0 * x0).sum())
        del x0
    assert abs(energy - float(np.sum(spr.S_ ** 2))) <= 1e-10 * energy
    rec = spr.reconstruct(spr.Ar[:1], to_host=False)
    assert tuple(rec.shape) == (1, n_loc)
    tail = np.sqrt(np.sum(spr.S_[r:] ** 2))
    sc = spr._d['scale']
    d = t.cat([(rec[0, :cut] - Xd[:cut, 0]) / sc[feats[0]], (rec[0, cut:] - Xd[cut:, 0]) / sc[feats[1]]])
    assert float(t.linalg.norm(d)) <= 1.05 * tail
    spr.optimal_placement()
    p1 = spr.sensors_.copy()
    spr.optimal_placement()
    np.testing.assert_array_equal(p1, spr.sensors_)
    assert len(set(p1.tolist())) == r and p1.min() >= row0 and p1.max() < row0 + n_loc and spr.pivot_gap_.min() > 1e-9
    _slab_pivot_check(eng, Ur, r)
    del Xd, spr, rec, d
    t.cuda.empty_cache()


def test_properties_at_config5_share(eng):
    """BASELINE config 5 as ONE GPU of eight holds it, at full size: 6.25M cells x 16 features x 512 snapshots in f32
    storage (100M rows, 204.8 GB) + f32 basis (51.2 GB), 128 modes: finite spectrum, orthonormal basis (to f32 rounding),
    energy identity, best-rank-r residual of a training column, reproducible distinct sensors whose pivot gaps are far
    above f32 rounding, slab-vs-dgeqp3 -- checked in 5M-row slices."""
    import bench
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
    from openmeasure_amd.synth import make_R
    t = eng.torch
    t.cuda.empty_cache()
    free, _ = t.cuda.mem_get_info()
    if free < 275e9:
        pytest.skip(f'needs 275 GB of free HBM, {free / 1e9:.0f} GB available')
    wl = bench.WORKLOADS['c5']
    n_points, F, m, r = wl['cells'], wl['features'], wl['m'], wl['s']
    n = n_points * F
    R = eng.to_device(make_R(m, r, seed=1234))
    Xd = eng.synth(n, m, 0, n_points, R, 1e-3, 1234, dtype=t.float32)
    spr = SPR(DeviceMatrix(Xd, basis='f32'), F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    assert np.isfinite(spr.S_).all() and spr.S_[0] / spr.S_[r - 1] < 1e4
    Ur, mu, inv = spr._d['Ur'], spr._d['rowmean'], spr._d['inv_scale']
    assert Ur.dtype == t.float32 and tuple(Ur.shape) == (n, r)
    step = 1_250_000                                                     # divides n_points: slices never straddle a feature
    gram_u = t.zeros((r, r), dtype=t.float64, device=Ur.device)
    energy = 0.0
    for i0 in range(0, n, step):
        u = Ur[i0:i0 + step].double()
        gram_u += u.T @ u
        x0 = (Xd[i0:i0 + step].double() - mu[i0:i0 + step, None]) * inv[i0 // n_points]
        energy += float((x0 * x0).sum())
        del x0, u
    assert np.abs(eng.to_host(gram_u) - np.eye(r)).max() < 5e-6          # f32-rounded entries, 1e8 rows
    assert abs(energy - float(np.sum(spr.S_ ** 2))) <= 1e-9 * energy
    rec = spr.reconstruct(spr.Ar[:1], to_host=False)
    tail = np.sqrt(np.sum(spr.S_[r:] ** 2))
    err2 = 0.0
    for i0 in range(0, n, 4 * step):
        dd = (rec[0, i0:i0 + 4 * step] - Xd[i0:i0 + 4 * step, 0].double()) * inv[i0 // n_points]
        err2 += float((dd * dd).sum())
        del dd
    assert np.sqrt(err2) <= 1.05 * tail + 1e-6 * np.sqrt(energy)       # + the f32 rounding of the stored basis
    spr.optimal_placement()
    p1 = spr.sensors_.copy()
    spr.optimal_placement()
    np.testing.assert_array_equal(p1, spr.sensors_)
    assert len(set(p1.tolist())) == r
    assert spr.pivot_gap_.min() > 1e-6, spr.pivot_gap_.min()            # far above f32 rounding of the basis (6e-8)
    _slab_pivot_check(eng, Ur, r, rows=1_000_000)
    del Xd, spr, rec
    t.cuda.empty_cache()


@pytest.mark.parametrize('dtype', ['f64', 'f32'])
@pytest.mark.parametrize('scale_type,axis_cnt', [('std', None), ('pareto', 1), ('range', None), ('median', 1), ('l2-norm', 1),
                                                 ('max', None), ('level', 1)])
@pytest.mark.parametrize('n_points,F,m,r', [(700, 3, 40, 9), (300, 4, 300, 12), (200, 2, 600, 10)])
def test_option_matrix_vs_oracle(eng, dtype, scale_type, axis_cnt, n_points, F, m, r):
    """Scalings x centring modes x storage precision x (narrow | column-split wide) snapshot counts, end to end
    against the oracle on the same stored values: statistics tight, sensors of the stored basis exact, fields 1e-6."""
    from openmeasure_amd.sparse_sensing import SPR
    X = synth_host(n_points, F, m, min(m, 2 * r), 0.75, 1e-3, 91) * 0.05 + 5.0       # positive: 'level' etc. well defined
    if dtype == 'f32':
        X = X.astype(np.float32)
    Xw = X.astype(np.float64)
    n = n_points * F
    st = orc.fit(Xw, F, 'number', r, scale_type=scale_type, axis_cnt=axis_cnt)
    spr = SPR(X, F, None, engine=eng)
    spr.fit(scale_type=scale_type, axis_cnt=axis_cnt, select_modes='number', n_modes=r)
    np.testing.assert_allclose(spr.X_cnt, st['X_cnt'], rtol=1e-12, atol=1e-12 * np.abs(st['X_cnt']).max())
    np.testing.assert_allclose(spr.X_scl, st['X_scl'], rtol=1e-10)
    np.testing.assert_allclose(spr.Sigma_r, st['Sigma_r'], rtol=1e-7)
    C = spr.optimal_placement()
    want, _ = orc.qr_pivots(spr.Ur.astype(np.float64))
    np.testing.assert_array_equal(spr.sensors_, want)
    spr.train(C)
    y = np.zeros((r, 3)); y[:, 0] = Xw[spr.sensors_, 1]; y[:, 2] = spr.sensors_ // n_points
    a, _ = spr.predict(y)
    x_rec = spr.reconstruct(a)
    # the training column is reproduced up to the truncation error of the rank-r basis
    x0_err = (x_rec[:, 0] - Xw[:, 1]) / st['X_scl'][:, 0]
    tail = np.sqrt(np.sum(st['S'][r:] ** 2))
    assert np.linalg.norm(x0_err) <= 1.5 * tail + 1e-6 * np.linalg.norm((Xw[:, 1] - st['X_cnt'][:, 0]) / st['X_scl'][:, 0])
    # and agrees with the oracle's reconstruction from the same sensors
    sg = align_signs(spr.Ar, st['Ar'])
    Cd = np.zeros((r, n)); Cd[np.arange(r), spr.sensors_] = 1.0
    Ur_ref = st['Ur'] * sg
    A_ref, _ = orc.predict_ols([y], orc.train_theta(Cd, Ur_ref, n), Cd, st['X_cnt'], st['X_scl'], n_points)
    assert rel_fro(x_rec, orc.reconstruct(A_ref, Ur_ref, st['X_cnt'], st['X_scl'])) <= REL_FRO


@pytest.mark.parametrize('n_points,F,m,select,n_modes,k', [(500, 3, 600, 'number', 40, 80), (400, 2, 300, 'number', 200, 200),
                                                         (300, 2, 1024, 'number', 16, 32), (260, 2, 520, 'variance', 99.9, 24)])
def test_wide_shapes_end_to_end_vs_oracle(eng, n_points, F, m, select, n_modes, k):
    """Shapes beyond one launch -- m > 512 (Gram as slice pairs, streamed-W projection) and r > 128 (column groups in
    projection / measure / reconstruct, the wide sweep and solve kernels): fit -> optimal_placement -> train -> predict
    -> reconstruct against the oracle; sensors exact and ordered, field 1e-6.  The reference accepts any m and any
    r <= m (:272-279, :336, :739)."""
    from openmeasure_amd.sparse_sensing import SPR
    rho = 10 ** (-2.0 / max(k - 1, 1))                          # k designed modes over two decades, noise floor far below
    X = synth_host(n_points, F, m, k, rho, 1e-6, 4000 + m)
    n = n_points * F
    st = orc.fit(X, F, select, n_modes)
    r = st['r']
    spr = SPR(X, F, None, engine=eng)
    spr.fit(select_modes=select, n_modes=n_modes)
    assert spr.r == r and spr.Ur.shape == (n, r)
    np.testing.assert_allclose(spr.Sigma_r, st['Sigma_r'], rtol=1e-8)
    # the subspace (the columns themselves are only defined up to rotations inside clusters of close singular values)
    z = np.random.default_rng(1).standard_normal(n)
    assert rel_fro(spr.Ur @ (spr.Ur.T @ z), st['Ur'] @ (st['Ur'].T @ z)) < 1e-8
    C = spr.optimal_placement()
    piv, _ = orc.qr_pivots(st['Ur'])
    np.testing.assert_array_equal(spr.sensors_, piv)
    assert C.shape == (r, n)
    spr.train(C)
    ys = []
    for j in (0, 1):
        y = np.zeros((r, 3)); y[:, 0] = X[piv, j] + (1e-3 if j else 0.0); y[:, 2] = piv // n_points
        if j == 1:
            y[:, 1] = 0.01 * (1.0 + np.arange(r) % 3)
        ys.append(y)
    A, As = spr.predict(ys)
    Cd = np.zeros((r, n)); Cd[np.arange(r), piv] = 1.0
    Theta_ref = orc.train_theta(Cd, st['Ur'], n)
    A_ref, _ = orc.predict_ols(ys, Theta_ref, Cd, st['X_cnt'], st['X_scl'], n_points)
    X_ref = orc.reconstruct(A_ref, st['Ur'], st['X_cnt'], st['X_scl'])
    X_rec = spr.reconstruct(A)
    assert X_rec.shape == (n, 2) and rel_fro(X_rec, X_ref) <= REL_FRO


def test_all_modes_of_a_wide_matrix(eng):
    """fit(select_modes='variance', n_modes=100) -- the option of the reference's own tests/test_rom.py:48-55 -- at m = 300:
    every mode kept (r = m = 300, among them the null mode of the row centring), training snapshots reproduced through
    reconstruct (tests/test_rom.py:82-85) and through placement -> train -> predict."""
    from openmeasure_amd.sparse_sensing import SPR
    n_points, F, m = 300, 2, 300
    X = synth_host(n_points, F, m, 60, 0.9, 1e-2, 4242)
    n = n_points * F
    spr = SPR(X, F, None, engine=eng)
    spr.fit(select_modes='variance', n_modes=100)
    assert spr.r == m and spr.Ur.shape == (n, m) and spr.Ar.shape == (m, m)
    X_rec = spr.reconstruct(spr.Ar[:3])
    assert rel_fro(X_rec, X[:, :3]) <= REL_FRO
    C = spr.optimal_placement()
    assert C.shape == (m, n) and len(set(spr.sensors_.tolist())) == m
    spr.train(C)
    y = np.zeros((m, 3)); y[:, 0] = X[spr.sensors_, 5]; y[:, 2] = spr.sensors_ // n_points
    a, _ = spr.predict(y)
    assert rel_fro(spr.reconstruct(a)[:, 0], X[:, 5]) <= 1e-5


@pytest.mark.parametrize('name', ['g2_num4', 'g3_num16', 'f32_g3_num8'])
def test_documented_idioms_on_one_hot_rows(eng, name, monkeypatch):   # README.md:160-184 / INTEGRATION.md on both return types
    from tests.conftest import load_golden
    run_documented_idioms(load_golden(name), eng, monkeypatch)


def test_one_hot_rows_returned_at_scale(eng):
    """Above the dense limit optimal_placement hands back a OneHotRows (8 M rows x 32 sensors = 2 GB dense): the documented
    idioms run on it without ever building the dense matrix, and train() takes it."""
    import torch
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix, OneHotRows
    from openmeasure_amd.synth import make_R
    n_points, F, m, s_ = 2_000_000, 4, 64, 32
    R = eng.to_device(make_R(m, s_, seed=5))
    Xd = eng.synth(n_points * F, m, 0, n_points, R, 1e-3, 5)
    spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=s_)
    C = spr.optimal_placement()
    assert isinstance(C, OneHotRows) and C.shape == (s_, n_points * F)
    rows = np.argmax(C, axis=1)
    np.testing.assert_array_equal(rows, spr.sensors_)
    assert all(np.argmax(C[i, :]) == rows[i] for i in range(s_))
    x = eng.to_host(Xd[:, 3])
    y = np.zeros((s_, 3)); y[:, 0] = C @ x; y[:, 2] = rows // n_points
    np.testing.assert_array_equal(y[:, 0], x[rows])
    spr.train(C)
    a, _ = spr.predict(y)
    xr = spr.reconstruct(a)
    assert rel_fro(xr[:, 0], x) < 1e-2                              # training column: rank-32 truncation error only
    with pytest.raises(MemoryError):
        OneHotRows(rows, 10 ** 10).toarray()
    del Xd, spr
    torch.cuda.empty_cache()


def test_state_read_before_fit_raises_attribute_error(eng):       # reference: the attribute does not exist yet
    from openmeasure_amd.sparse_sensing import SPR
    X = np.random.default_rng(0).random((40, 6))
    spr = SPR(X, 2, None, engine=eng)
    for attr in ('Ur', 'X_cnt', 'X_scl', 'X0', 'Ar', 'Sigma_r', 'Vr'):
        with pytest.raises(AttributeError):
            getattr(spr, attr)
    for call in (spr.optimal_placement, lambda: spr.reconstruct(np.ones(3)), lambda: spr.train(np.zeros((2, 40))),
                 lambda: spr.unscale_data(np.ones(40))):
        with pytest.raises(AttributeError):
            call()


@pytest.mark.parametrize('m', [12, 64, 300])
def test_constant_feature_and_nan_raise_linalgerror(eng, m):
    """A constant feature (X_scl = 0 -> X0 = 0/0, :169) and a NaN / Inf in X make np.linalg.svd raise LinAlgError at :272;
    so must fit(), on the sync-free device route (m = 12: statistics, spectrum and projection enqueued before the
    verdict comes back), on the host-spectrum route (m = 64) and on the wide path (m = 300)."""
    from openmeasure_amd.sparse_sensing import SPR
    rng = np.random.default_rng(m)
    X = rng.standard_normal((600, m))
    Xc = X.copy(); Xc[300:] = 3.0                                   # mean exact: X_scl = 0 exactly
    Xn = X.copy(); Xn[77, 3] = np.nan
    Xi = X.copy(); Xi[411, 0] = np.inf
    for bad in (Xc, Xn, Xi):
        spr = SPR(bad, 2, None, engine=eng)
        with pytest.raises(np.linalg.LinAlgError):
            with np.errstate(all='ignore'):
                spr.fit(select_modes='number', n_modes=3)
    ok = SPR(X, 2, None, engine=eng)                                # the same engine still works afterwards
    ok.fit(select_modes='number', n_modes=3)
    assert np.isfinite(ok.Sigma_r).all()


@pytest.mark.parametrize('m,r', [(12, 5), (20, 8)])
def test_device_fit_hands_over_without_second_read(eng, m, r):
    """Small m (sync-free device route) with sigma_1/sigma_r far above 1e4: the route's verdict comes back bad, the host
    route takes over from the Gram blocks already formed (refinement pass) and the sensors are the reference's."""
    from openmeasure_amd.sparse_sensing import SPR
    X = synth_host(2000, 3, m, m, 10 ** (-7.0 / (r - 1)), 1e-15, 600 + m)
    st = orc.fit(X, 3, 'number', r)
    assert st['Sigma_r'][0] / st['Sigma_r'][-1] > 1e6
    spr = SPR(X, 3, None, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    assert spr.gram_refine_passes_ >= 1 and getattr(spr, '_device_fit_fallback_', False)
    spr.optimal_placement()
    np.testing.assert_array_equal(spr.sensors_, orc.qr_pivots(st['Ur'])[0])


def test_partial_row_group_with_absent_features(eng):
    """RowShard(partial=True) -- one rank's block of a larger job run alone (bench.py --share-of): features without rows
    in the block take no part (scale 1, no Gram contribution) instead of poisoning the Gram matrix with 0/0; the fit
    equals the fit of the same rows as a stand-alone matrix of the present features."""
    from openmeasure_amd.sparse_sensing import SPR, RowShard
    n_points, F, m, r = 900, 4, 40, 6
    X = synth_host(n_points, F, m, 12, 0.7, 1e-3, 31)
    row0, n_loc = 300, 1200                                         # rows of features 0 (tail) and 1 (head + most): 2, 3 absent
    blk = np.ascontiguousarray(X[row0:row0 + n_loc])
    spr = SPR(blk, F, None, shard=RowShard(row0, n_points * F, partial=True, force_collectives=False), engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    assert np.isfinite(spr.S_).all() and np.isfinite(spr.Ur).all()
    np.testing.assert_array_equal(spr._scl_f[2:], 1.0)
    # oracle on the same rows: feature 0's rows and feature 1's rows as two blocks of their own
    a, b = blk[:600], blk[600:]
    Xc = np.vstack([(a - a.mean(1, keepdims=True)) / np.std(a), (b - b.mean(1, keepdims=True)) / np.std(b)])
    S_ref = np.linalg.svd(Xc, compute_uv=False)
    np.testing.assert_allclose(spr.S_[:r], S_ref[:r], rtol=1e-9)


@pytest.mark.parametrize('s_,r,cond', [(200, 200, 10.0), (450, 300, 1e3), (129, 129, 5.0), (1100, 1024, 30.0)])
def test_solve_ols_wide_vs_oracle(eng, s_, r, cond):
    """Normal equations + Cholesky + one refinement step for r > 128 (spr_solve_ols_wide_f64: matrices in a workspace):
    weighted and unweighted right-hand sides against the oracle's SVD pseudo-inverse (:873-878)."""
    rng = np.random.default_rng(s_ + r)
    Uq, _ = np.linalg.qr(rng.standard_normal((s_, r)))
    Vq, _ = np.linalg.qr(rng.standard_normal((r, r)))
    Theta = (Uq * np.logspace(0, -np.log10(cond), r)) @ Vq.T
    a_true = rng.standard_normal(r)
    ys = []
    for weighted in (False, True):
        y = np.zeros((s_, 3))
        y[:, 0] = Theta @ a_true + 1e-3 * rng.standard_normal(s_)
        if weighted:
            y[:, 1] = 0.05 * (1.0 + rng.random(s_))
        ys.append(y)
    Ar, As, y0, info = eng.solve_ols(eng.to_device(Theta), eng.to_device(np.zeros(s_)), eng.to_device(np.ones(1)),
                                     eng.to_device(np.stack(ys)))
    Ar, As, info = eng.to_host(Ar), eng.to_host(As), eng.to_host(info)
    assert not info[:, 0].any() and np.all(info[:, 1] < 1e13)
    A_ref, S_ref = _identity_problem(Theta, ys)
    for k in range(2):
        assert np.linalg.norm(Ar[k] - A_ref[k]) <= 1e-9 * cond * np.linalg.norm(A_ref[k]), (k, info[k])
    assert np.linalg.norm(As[1] - S_ref[1]) <= 1e-9 * cond * np.linalg.norm(S_ref[1]) and not As[0].any()
    np.testing.assert_allclose(eng.to_host(y0)[1, :, 1], ys[1][:, 1])


@pytest.mark.parametrize('m,r', [(256, 64), (512, 96), (300, 40)])
def test_gap_filler_changes_nothing_but_the_clock(eng, monkeypatch, m, r):
    """ROM.gap_filler: from the second fit() on, the Gram kernel is queued once more (results discarded) into the host
    gap between the Gram pass and the projection.  Every fitted quantity is bit for bit what a fit without the filler
    gives, and the top-r eigen route (m >= 96) gives the sensors of the oracle."""
    from openmeasure_amd.sparse_sensing import SPR
    n_points, F = (30_000 if m == 256 else 24_000), 3          # (a wide X is filled with the 256-column kernel on its first slice)
    X = synth_host(n_points, F, m, 100, 0.93, 1e-3, 11)
    monkeypatch.setattr(SPR, '_GAP_FILL_MIN_MS', 0.0)          # whatever this host's eigen-solve takes, fill its gap
    assert SPR.gap_filler is False                             # opt-in (round 5): a plain fit() queues no discarded work
    a = SPR(X, F, None, engine=eng)
    a.gap_filler = True
    fills = []
    for _ in range(4):
        a.fit(select_modes='number', n_modes=r)
        fills.append(a._gap_fill_rows)
    assert fills[0] == 0 and fills[-1] >= 65536, fills      # no history in the first call; later ones fill the gap
    b = SPR(X, F, None, engine=eng)
    for _ in range(2):
        b.fit(select_modes='number', n_modes=r)
    assert not hasattr(b, '_gap_fill_rows')
    monkeypatch.setenv('SPR_GAP_FILLER', '0')                  # the environment switch overrides the attribute
    a.fit(select_modes='number', n_modes=r)
    monkeypatch.delenv('SPR_GAP_FILLER')
    for name in ('Ur', 'X_cnt', 'Sigma_r', 'Ar'):
        np.testing.assert_array_equal(getattr(a, name), getattr(b, name))
    a.optimal_placement(); b.optimal_placement()
    np.testing.assert_array_equal(a.sensors_, b.sensors_)
    st = orc.fit(X, F, select_modes='number', n_modes=r)
    piv, _ = orc.qr_pivots(st['Ur'])
    np.testing.assert_array_equal(a.sensors_, piv)
    np.testing.assert_allclose(a.Sigma_r, st['Sigma_r'], rtol=1e-8)


@pytest.mark.parametrize('axis_cnt,m,r', [(1, 24, 6), (None, 24, 6), (1, 256, 32), (None, 256, 32), (1, 300, 20)])
def test_fit_precentres_large_offsets(eng, axis_cnt, m, r):
    """Fit-level check of the pre-centring safeguard: data whose centre is 1e6-1e7 times its fluctuation.  fit()
    must choose the pre-centred projection on its own (precentered_), for row centring and for scalar centring, on the
    device-spectrum route (m <= 24), the host route and the wide route; the basis then agrees with a fit of the SAME
    fluctuation without the offset -- for which the epilogue form is exact -- far better than the epilogue form would."""
    from openmeasure_amd.sparse_sensing import SPR
    rng = np.random.default_rng(3 * m + (axis_cnt or 0))
    n_points, F = 6000, 1               # one feature: the block scale is then a common factor that Ur = X0 V / S does not see
    n = n_points * F
    fl = rng.standard_normal((n, 12)) @ ((0.7 ** np.arange(12))[:, None] * rng.standard_normal((12, m))) + 1e-3 * rng.standard_normal((n, m))
    if axis_cnt == 1:
        fl = fl - fl.mean(axis=1, keepdims=True)
        X = 1e6 * (1.0 + rng.random((n, 1))) + fl
    else:
        X = 1e7 + fl
    a = SPR(np.ascontiguousarray(X), F, None, engine=eng)
    a.fit(axis_cnt=axis_cnt, select_modes='number', n_modes=r)
    assert a.precentered_ is True
    b = SPR(np.ascontiguousarray(fl), F, None, engine=eng)              # offset removed before the data is handed over
    b.fit(axis_cnt=axis_cnt, select_modes='number', n_modes=r)
    assert b.precentered_ is False
    # X = offset + fl is rounded to f64 at magnitude 1e6-1e7: the data themselves differ by 1e-10 relative to the fluctuation,
    # amplified by sigma_1/sigma_i in the small modes
    kappa = b.Sigma_r[0] / b.Sigma_r
    sg = np.sign(np.sum(a.Ur * b.Ur, axis=0))
    err = np.abs(a.Ur * sg - b.Ur).max(axis=0) / np.abs(b.Ur).max()
    assert np.all(err <= 2e-9 * kappa), (err / kappa).max()
    np.testing.assert_allclose(a.Sigma_r / a.Sigma_r[0], b.Sigma_r / b.Sigma_r[0], rtol=1e-7)   # up to the common block scale


@pytest.mark.parametrize('r,s_extra', [(4, 0), (16, 8), (40, 0), (200, 0)])
def test_predict_failure_modes_on_the_device(eng, r, s_extra):
    """The reference's failure modes of predict() (:868-878), sent through solve.hip / solve_pinv.hip (and their wide twins):
    uncertainties that are zero for SOME sensors make W = diag(1/0) (:872) and np.linalg.pinv raises LinAlgError; a NaN
    measurement raises nothing and returns NaN coefficients; an infinite measurement likewise propagates."""
    from openmeasure_amd.sparse_sensing import SPR
    rng = np.random.default_rng(r)
    n_points, F, m = 1500, 2, max(2 * r, 24)
    X = synth_host(n_points, F, m, min(m, r + 8), 0.9, 1e-3, 50 + r)
    spr = SPR(X, F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    C = spr.optimal_placement()
    piv = spr.sensors_
    if s_extra:                                                       # more sensors than modes: a general measurement matrix
        more = rng.choice(np.setdiff1d(np.arange(X.shape[0]), piv), s_extra, replace=False)
        piv = np.concatenate([piv, more])
        import scipy.sparse as sp
        C = sp.csr_matrix((np.ones(len(piv)), (np.arange(len(piv)), piv)), shape=(len(piv), X.shape[0]))
    spr.train(C)
    s = len(piv)
    y = np.zeros((s, 3)); y[:, 0] = X[piv, 1]; y[:, 2] = piv // n_points
    y[:, 1] = 0.01 * (1 + rng.random(s))
    a_ok, sig_ok = spr.predict(y)
    assert np.all(np.isfinite(a_ok)) and np.all(np.isfinite(sig_ok))
    # (i) partially-zero uncertainties: 1/0 in W -> LinAlgError (the reference: SVD did not converge, :873)
    ybad = y.copy(); ybad[s // 2, 1] = 0.0
    with pytest.raises(np.linalg.LinAlgError):
        spr.predict(ybad)
    # ... also inside a batch whose other vectors are fine
    with pytest.raises(np.linalg.LinAlgError):
        spr.predict([y, ybad, y])
    # (ii) a NaN measurement: no exception in the reference (pinv sees only W Theta), NaN coefficients
    ynan = y.copy(); ynan[0, 0] = np.nan
    a_nan, sig_nan = spr.predict(ynan)
    assert np.all(np.isnan(a_nan)) and np.all(np.isfinite(sig_nan))
    np.testing.assert_allclose(sig_nan, sig_ok, rtol=1e-9)
    # (iii) a NaN uncertainty poisons W Theta itself: pinv raises
    ynu = y.copy(); ynu[1, 1] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        spr.predict(ynu)
    # the object is still usable afterwards
    a_again, _ = spr.predict(y)
    np.testing.assert_array_equal(a_again, a_ok)


@pytest.mark.parametrize('n,r,n_p', [(9_000_123, 8, 1), (3_100_001, 16, 3)])
def test_reconstruct_to_host_in_chunks(eng, n, r, n_p):
    """The reference's output contract (:371-375: a host (n, n_p) ndarray) for fields above 64 MiB: row chunks whose D2H
    copies overlap the next chunk's kernel, landing in page-locked memory -- bit for bit the field the one-launch path
    leaves in HBM, features straddling the chunk boundaries."""
    import torch
    from openmeasure_amd.sparse_sensing import SPR
    g = torch.Generator(device='cuda').manual_seed(n)
    F = 3
    n_points = n // F
    n = n_points * F
    Ur = torch.randn((n, r), generator=g, dtype=torch.float64, device='cuda')
    rowmean = torch.randn((n,), generator=g, dtype=torch.float64, device='cuda')
    scale = eng.to_device(np.array([1.5, 0.25, 3.0]))
    A = eng.to_device(np.random.default_rng(r).standard_normal((n_p, r)))
    dev = eng.reconstruct(Ur, 0, n_points, F, rowmean, scale, A)
    host = eng.reconstruct_to_host(Ur, 0, n_points, F, rowmean, scale, A, chunks=5)
    assert host is not None and host.shape == (n_p, n)
    np.testing.assert_array_equal(host, dev.cpu().numpy())
    # through the class: reconstruct() returns the (n, n_p) array of the reference
    spr = SPR(np.zeros((F * 4, 2)), F, None, engine=eng)
    spr.__dict__.update(_n_global=n, n_points=n_points, r=r)
    spr._d.update(Ur=Ur, rowmean=rowmean, scale=scale)
    out = spr.reconstruct(eng.to_host(A))
    assert out.shape == (n, n_p)
    np.testing.assert_array_equal(out, host.T)
    # small results keep the one-launch path
    assert eng.reconstruct_to_host(Ur[:1000], 0, n_points, F, rowmean[:1000], scale, A) is None


def test_field_unstage_kernel(eng):
    """spr_field_unstage_f64: the staged blocks of a sharded multi-vector reconstruct() -> the vectors side by side."""
    import torch
    for world, n_p, n_loc in ((2, 3, 1001), (4, 2, 4096), (3, 5, 7)):
        st = torch.randn((world, n_p, n_loc), dtype=torch.float64, device='cuda')
        out = eng.field_unstage(st)
        assert torch.equal(out, st.permute(1, 0, 2).reshape(n_p, world * n_loc))
        host = eng.stage_to_host(st)
        np.testing.assert_array_equal(host, out.cpu().numpy())


def test_pinned_result_budget(eng, monkeypatch):
    """Host results above 4 MiB are page-locked memory; the ones the caller still holds are counted against a budget and a
    result beyond it falls back to the pageable copy -- same values either way."""
    import torch
    t = torch.arange(10_000_000, dtype=torch.float64, device='cuda')          # 80 MB
    monkeypatch.setenv('SPR_PINNED_RESULT_GB', '0.1')
    eng.__dict__.pop('_pinned_budget', None)                                  # (the variable is read once per engine)
    a = eng.to_host(t)
    assert len([1 for w, _ in eng._pinned_live if w() is not None]) >= 1
    eng.__dict__.pop('_pinned_warned', None)
    with pytest.warns(RuntimeWarning, match='page-locked results are still referenced'):      # said once, not silently (round 6)
        b = eng.to_host(t)                                                    # 160 MB alive > 0.1 GiB: pageable
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        b2 = eng.to_host(t)                                                   # ... and only once per engine
    np.testing.assert_array_equal(b, b2)
    del b2
    n_live = len([1 for w, _ in eng._pinned_live if w() is not None])
    np.testing.assert_array_equal(a, b)
    del a
    c = eng.to_host(t)                                                        # the first one was released: pinned again
    assert len([1 for w, _ in eng._pinned_live if w() is not None]) == n_live
    np.testing.assert_array_equal(b, c)
    monkeypatch.setenv('SPR_PINNED_RESULT_GB', '0')
    eng.__dict__.pop('_pinned_budget', None)
    before = len(eng._pinned_live)
    d = eng.to_host(t)
    assert len([1 for w, _ in eng._pinned_live if w() is not None]) <= before
    np.testing.assert_array_equal(d, c)
    eng.__dict__.pop('_pinned_budget', None)                                  # the next to_host() reads the restored environment


@pytest.mark.parametrize('switch', ['SPR_PROJECT_WS=0', 'SPR_QR_FUSED_STEPS=0', 'SPR_QR_ORTH_TILE=0', 'SPR_QR_EPOCH_ILP=0',
                                    'SPR_QR_EPOCH_ILP=2', 'SPR_QR_DIRECT=0', 'SPR_RECONSTRUCT_DIRECT=0', 'SPR_RECONSTRUCT_DIRECT=3',
                                    'SPR_DL_KERNEL=0', 'SPR_PROJECT_STREAM=1', 'SPR_WS_WG_PER_CU=4'])
def test_ab_switches_keep_parity(switch):
    """The A/B switches of the library and of the Python layer (include/spr_hip.h) select kernel forms that the default path does not
    take at these shapes; the library reads them once per process, so each runs tests/_switch_check.py in a process of its own: the
    whole path on five shapes against the oracle, sensors exact."""
    import subprocess
    import sys
    key, val = switch.split('=')
    env = dict(os.environ)
    env[key] = val
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tests', '_switch_check.py')], env=env, cwd=root, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and 'SWITCH_CHECK_OK' in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_engine_close_and_reuse(eng):
    """HipEngine.close() (also registered with atexit): staging buffers, copy threads and cached page-locked blocks go back;
    host arrays handed out before stay valid (they own their memory) and the engine makes its buffers again on demand."""
    import torch
    t = torch.arange(2_000_000, dtype=torch.float64, device='cuda')           # 16 MB: a page-locked result
    small = torch.arange(1000, dtype=torch.float64, device='cuda')
    a = eng.to_host(t, result=True)
    assert any(w() is a for w, _ in eng._pinned_live)                          # handed out as page-locked memory of its own
    s0 = eng.to_host(small)
    w = eng.upload_reuse(('test', 0), np.arange(12.0).reshape(3, 4))
    assert torch.equal(w.cpu(), torch.arange(12.0, dtype=torch.float64).reshape(3, 4))
    big = torch.randn((2_000_003, 5), dtype=torch.float64, device='cuda')
    ref = big.cpu().numpy()
    np.testing.assert_array_equal(eng._to_host_staged(big), ref)               # starts the copy threads
    eng.close()
    assert eng.__dict__.get('_copy_pool') is None and '_dstage_slots' not in eng.__dict__ and eng._stage is None
    np.testing.assert_array_equal(a, np.arange(2_000_000, dtype=np.float64))   # still readable after the engine let go
    eng.close()                                                                # idempotent
    np.testing.assert_array_equal(eng.to_host(t, result=True), a)
    np.testing.assert_array_equal(eng.to_host(t), a)                           # an internal download: through the shared stage
    np.testing.assert_array_equal(eng.to_host(small), s0)
    np.testing.assert_array_equal(eng._to_host_staged(big), ref)
    w = eng.upload_reuse(('test', 0), np.arange(12.0).reshape(3, 4) + 1)
    assert torch.equal(w.cpu(), torch.arange(12.0, dtype=torch.float64).reshape(3, 4) + 1)
    X = synth_host(3000, 2, 24, 8, 0.8, 1e-3, 3)
    from openmeasure_amd.sparse_sensing import SPR
    spr = SPR(X, 2, None, engine=eng)
    spr.fit(select_modes='number', n_modes=6)
    eng.close()                                                                # between two calls of a fitted object
    spr.optimal_placement()
    np.testing.assert_array_equal(spr.sensors_, orc.qr_pivots(orc.fit(X, 2, 'number', 6)['Ur'])[0])


def test_big_attributes_stream_to_the_host(eng, monkeypatch):
    """t1: `spr.Ur` / `spr.X0` stay obtainable as host ndarrays at sizes that can neither be page-locked as a whole nor be
    materialised next to X in HBM: staged D2H through two pinned buffers, X0 by row blocks -- same values as the one-shot paths."""
    import torch
    from openmeasure_amd.sparse_sensing import SPR
    try:
        _big_attributes_body(eng, monkeypatch, torch, SPR)
    finally:
        eng.__dict__.pop('_pinned_budget', None)                               # later tests read the restored environment


def _big_attributes_body(eng, monkeypatch, torch, SPR):
    t = torch.randn((3_000_017, 7), dtype=torch.float64, device='cuda')       # 168 MB, ragged against the 64 MiB chunks
    ref = t.cpu().numpy()
    np.testing.assert_array_equal(eng._to_host_staged(t), ref)
    monkeypatch.setenv('SPR_PINNED_RESULT_GB', '0')                            # beyond the budget: to_host() takes the staged path
    eng.__dict__.pop('_pinned_budget', None)
    np.testing.assert_array_equal(eng.to_host(t), ref)
    t32 = t[:1_000_003].float()
    np.testing.assert_array_equal(eng._to_host_staged(t32), t32.cpu().numpy())
    X = synth_host(40_000, 3, 24, 8, 0.8, 1e-3, 9)
    a = SPR(X, 3, None, engine=eng)
    a.fit(select_modes='number', n_modes=4)
    whole = a.X0
    b = SPR(X, 3, None, engine=eng)
    monkeypatch.setattr(SPR, '_X0_BLOCK_BYTES', 24 * 8 * 10_007)               # blocks that end inside features
    b.fit(select_modes='number', n_modes=4)
    np.testing.assert_array_equal(b.X0, whole)
    np.testing.assert_allclose(whole, orc.scale_data_std(X, 3)[2], rtol=0, atol=1e-12)


@pytest.mark.parametrize('n_points,F,m,q,f32,pre,i0', [(5000, 3, 256, 256, False, False, 0), (3333, 2, 256, 200, False, True, 128),
                                                       (4000, 1, 512, 256, True, False, 0), (2000, 2, 300, 300, False, False, 16),
                                                       (2500, 3, 130, 129, False, True, 0), (1800, 2, 256, 256, True, True, 64)])
def test_project_f64_sixteen_tile_form(eng, n_points, F, m, q, f32, pre, i0):
    """engine.project_f64 (what fit()'s refinement pass projects with): up to 256 columns in ONE launch of the streamed-W kernel's
    16-tile form (round 4: X read once per pass), wider W in groups of 256, unaligned / m % 4 != 0 shapes in groups of 128 --
    against ((X - mean) W) / X_scl in NumPy, row blocks starting inside a feature."""
    import torch
    rng = np.random.default_rng(m + q + i0)
    n = n_points * F
    X = rng.standard_normal((n, m)) * 2.0 + rng.standard_normal((n, 1)) * 5.0
    if f32:
        X = X.astype(np.float32).astype(np.float64)
    mu = X.mean(axis=1)
    scl = 0.5 + rng.random(F)
    W = rng.standard_normal((m, q))
    Xd = eng.to_device(X.astype(np.float32), dtype=torch.float32) if f32 else eng.to_device(X)
    rows = n - i0 - 7
    out = eng.empty((rows, q + (q & 1)))
    eng.project_f64(Xd, i0, rows, 0, n_points, F, eng.to_device(1.0 / scl), eng.to_device(W), eng.to_device(mu), out, center=True,
                    precenter=pre)
    feat = np.arange(i0, i0 + rows) // n_points
    ref = ((X[i0:i0 + rows] - mu[i0:i0 + rows, None]) @ W) / scl[feat][:, None]
    got = eng.to_host(out)[:, :q]
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    out0 = eng.empty((rows, q + (q & 1)))
    eng.project_f64(Xd, i0, rows, 0, n_points, F, eng.to_device(np.ones(F)), eng.to_device(W), None, out0, center=False)
    ref0 = X[i0:i0 + rows] @ W
    assert np.abs(eng.to_host(out0)[:, :q] - ref0).max() <= 1e-12 * np.abs(ref0).max()


def test_small_downloads_by_kernel_and_ticket(eng, monkeypatch):
    """round 5: HipEngine.to_host of small results goes through spr_download_bytes (a kernel writes into page-locked memory
    and raises a ticket the host polls) -- every dtype / shape the path carries, the `then` hook, values equal to the copy + event
    path, and the reusable W upload (spr_upload_bytes into a buffer that is overwritten by the next call with the same key)."""
    import torch
    from openmeasure_amd.engine import HipEngine
    rng = np.random.default_rng(3)
    cases = [rng.standard_normal((64, 64)), rng.standard_normal(7), rng.integers(-5, 5, size=(3, 5, 2)).astype(np.int64),
             rng.standard_normal((130, 2)).astype(np.float32), rng.integers(0, 255, size=(16,)).astype(np.uint8),
             rng.standard_normal((512, 256))]                                  # 1 MiB: the largest the kernel path takes
    assert eng._dl_kernel
    called = []
    for a in cases:
        t = torch.as_tensor(a).to(eng.device)
        got = eng.to_host(t, then=lambda: called.append(1))
        assert got.dtype == a.dtype and got.shape == a.shape
        np.testing.assert_array_equal(got, a)
        got[...] = 0                                                           # a fresh array every time, not the staging buffer
        np.testing.assert_array_equal(eng.to_host(t), a)
    assert len(called) == len(cases) and eng._dl[0]['seq'] >= 2 * len(cases)
    # a strided view and an odd byte count take the other path and still come back right
    t = torch.as_tensor(cases[0]).to(eng.device)
    np.testing.assert_array_equal(eng.to_host(t[:, ::2]), cases[0][:, ::2])
    np.testing.assert_array_equal(eng.to_host(torch.arange(5, dtype=torch.uint8, device=eng.device)), np.arange(5, dtype=np.uint8))
    other = HipEngine('cuda:0')
    other._dl_kernel = False
    for a in cases[:3]:
        np.testing.assert_array_equal(other.to_host(torch.as_tensor(a).to(eng.device)), a)
    W1, W2 = rng.standard_normal((64, 32)), rng.standard_normal((64, 32))
    d1 = eng.upload_reuse(('W', 1), W1)
    np.testing.assert_array_equal(eng.to_host(d1), W1)
    d2 = eng.upload_reuse(('W', 1), W2)
    assert d2.data_ptr() == d1.data_ptr()                                      # the same device buffer, new contents
    np.testing.assert_array_equal(eng.to_host(d2), W2)
    np.testing.assert_array_equal(eng.to_host(eng.upload_reuse(('W', 1), W1[:10])), W1[:10])   # another shape: a new buffer


def test_downloads_nest(eng):
    """A download issued from inside another download's `then` hook (fit()'s host gap launching a deferred reconstruct whose
    collective set-up downloads handles and verdicts -- ADVICE r05) gets a buffer, ticket and event of its own: both calls return
    their own payload, on the kernel + ticket path and on the copy + event path, two levels deep, sizes that differ."""
    import torch
    from openmeasure_amd.engine import HipEngine
    rng = np.random.default_rng(11)
    for e in (eng, HipEngine('cuda:0')):
        if e is not eng:
            e._dl_kernel = False                                               # the copy + event path (shared pinned stage)
        A, B, Cc = rng.standard_normal((256, 256)), rng.standard_normal((40,)), rng.standard_normal((3, 7, 8))
        big = rng.standard_normal((600_000,))                                  # 4.8 MB: the staged path on both engines
        tA, tB, tC, tbig = (torch.as_tensor(x).to(e.device) for x in (A, B, Cc, big))
        inner = {}

        def level2():
            inner['C'] = e.to_host(tC)
            inner['big2'] = e.to_host(tbig[:300_001])

        def level1():
            inner['B'] = e.to_host(tB, then=level2)
            inner['B2'] = e.to_host(tB * 2)                                    # the same depth used twice in one hook

        for _ in range(3):
            inner.clear()
            gotA = e.to_host(tA, then=level1)
            np.testing.assert_array_equal(gotA, A)
            np.testing.assert_array_equal(inner['B'], B)
            np.testing.assert_array_equal(inner['B2'], 2 * B)
            np.testing.assert_array_equal(inner['C'], Cc)
            np.testing.assert_array_equal(inner['big2'], big[:300_001])
            inner.clear()
            gotbig = e.to_host(tbig, then=level1)
            np.testing.assert_array_equal(gotbig, big)
            np.testing.assert_array_equal(inner['B'], B)
            np.testing.assert_array_equal(inner['C'], Cc)
        assert e._dl_depth == 0

        def boom():
            raise KeyError('hook')
        with pytest.raises(KeyError):
            e.to_host(tA, then=boom)
        assert e._dl_depth == 0                                                # the depth unwinds when a hook raises
        np.testing.assert_array_equal(e.to_host(tA), A)


def test_deferred_reconstruct_on_the_device(eng):
    """ROM.defer_reconstruct on the HIP engine: the launch recorded by reconstruct(wait=False) is enqueued in the next fit()'s
    host gap (behind the Gram download, before the projection overwrites the basis) with the basis it was called after."""
    from tests.parity import run_deferred_reconstruct
    run_deferred_reconstruct(eng)


def test_predict_outputs_in_one_download(eng):
    """round 6: the four outputs of the solve kernels (info, Ar, Ar_sigma, y0) are views of ONE buffer, each on a 256-byte boundary,
    and predict() brings them to the host in one download (HipEngine.to_host_views) -- same values as one download each, for odd r / s
    and several vectors, on both solve paths."""
    rng = np.random.default_rng(12)
    for s_, r, n_p in ((7, 5, 3), (64, 64, 1), (33, 9, 4)):
        Theta = rng.standard_normal((s_, r))
        Y = np.zeros((n_p, s_, 3))
        Y[:, :, 0] = rng.standard_normal((n_p, s_))
        Y[:, :, 1] = 0.1 + rng.random((n_p, s_))
        args = (eng.to_device(Theta), eng.to_device(np.zeros(s_)), eng.to_device(np.ones(1)), eng.to_device(Y))
        for solve in (eng.solve_ols, eng.solve_pinv):
            Ar, As, y0, info = solve(*args)
            assert all(t._base is Ar._base and t.data_ptr() % 256 == 0 for t in (Ar, As, y0, info))
            calls = []
            real = eng.to_host
            eng.to_host = lambda t, **kw: (calls.append(1), real(t, **kw))[1]
            try:
                got = eng.to_host_views(info, Ar, As, y0)
            finally:
                eng.to_host = real
            assert len(calls) == 1
            for a, t in zip(got, (info, Ar, As, y0)):
                np.testing.assert_array_equal(a, eng.to_host(t))
                assert a.shape == tuple(t.shape) and a.flags['OWNDATA']
    # tensors that do NOT share a buffer: one download each, same values
    a, b = eng.to_device(np.arange(6.0)), eng.to_device(np.arange(4.0))
    ga, gb = eng.to_host_views(a, b)
    np.testing.assert_array_equal(ga, np.arange(6.0))
    np.testing.assert_array_equal(gb, np.arange(4.0))


def test_integration_doc_binding_runs(eng, tmp_path):
    """INTEGRATION.md section 2 is a working binding, not prose: its two Python blocks (the ctypes stub with stats_gram, and the
    round-6 continuation with make_comm / sharded_first_pass) are executed as written -- only the library's path is filled in --
    and give the library's own results: the Gram blocks of the first, and the all-reduced, scaled Gram matrix of a one-rank
    sharded pass (the doc's route: communicator, spr_fit_gram_pass) equal to the unsharded steps' bit for bit."""
    import re
    import torch
    from openmeasure_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    stub = next(b for b in blocks if 'def stats_gram' in b)
    more = next(b for b in blocks if 'def sharded_first_pass' in b)
    ns = {}
    exec(stub.replace("C.CDLL('libspr_hip.so')", f"C.CDLL({_lib.LIB_PATH!r})"), ns)
    exec(more, ns)
    rng = np.random.default_rng(5)
    n_points, F, m = 3000, 3, 40
    X = rng.standard_normal((n_points * F, m)) + np.repeat(np.arange(F), n_points)[:, None]
    Xd = eng.to_device(X)
    torch.cuda.set_device(eng.device)
    rowmean, fstats, gram = ns['stats_gram'](Xd, n_points, F)
    rm, fs, gr = eng.stats_gram(Xd, 0, n_points, F)
    assert torch.equal(rowmean, rm) and torch.equal(gram, gr) and torch.equal(fstats, fs)
    comm = ns['make_comm'](0, 1, lambda b: b)
    try:
        rowmean2, G, scale, inv = ns['sharded_first_pass'](comm, 1, Xd, 0, n_points, F)
        packed, scale_ref, inv_ref = eng.gram_combine(gr, fs[None], 'std')
        torch.cuda.synchronize()
        assert torch.equal(rowmean2, rm) and torch.equal(G.reshape(-1), packed[:m * m]) and torch.equal(scale, scale_ref)
        assert torch.equal(inv, inv_ref)
    finally:
        assert eng.lib.spr_comm_destroy(comm) == 0


@pytest.mark.parametrize('m,f32', [(40, False), (256, True), (257, False), (300, False), (512, True)])
def test_fit_gram_pass_matches_the_separate_calls(eng, m, f32):
    """spr_fit_gram_pass (round 6: the first pass of fit() as ONE library call -- Gram kernel(s), finalize, [all-reduce], statistics
    merge; beyond 256 snapshots the column-split path with its three launches, BASELINE config 5's width) against the same steps
    called one by one through the engine: row means, Gram blocks, statistics slots, scaled Gram matrix and scales bit for bit;
    a row block that starts inside a feature; f64 and f32 storage."""
    import torch
    rng = np.random.default_rng(m)
    n_points, F, row0, n = 1500, 3, 700, 3300                     # rows 700 .. 3999 of 4500: three features, the first and last cut
    X = rng.standard_normal((n, 6)) @ rng.standard_normal((6, m)) + 0.05 * rng.standard_normal((n, m)) + 2.0
    Xd = eng.to_device(X.astype(np.float32) if f32 else X, dtype=torch.float32 if f32 else None)
    rowmean, buf, packed, scale, inv = eng.fit_gram_pass(Xd, row0, n_points, F, 'std', None, 1)
    rm, fs, gr = eng.stats_gram(Xd, row0, n_points, F)
    pk, sc, iv = eng.gram_combine(gr, fs[None], 'std')
    torch.cuda.synchronize()
    assert torch.equal(rowmean, rm)
    assert torch.equal(buf[:F * m * m].view(F, m, m), gr)
    assert torch.equal(buf[F * m * m:F * m * m + F * 3].view(F, 3), fs) and float(buf[-1]) == float(row0)
    assert torch.equal(packed, pk) and torch.equal(scale, sc) and torch.equal(inv, iv)
    assert bool(torch.isfinite(packed).all())
