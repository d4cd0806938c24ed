"""World-size-2 (and 3) tests of the row-sharded path over the gloo backend on CPU.

The HIP kernels cannot run here, so the local per-shard arithmetic is supplied by the NumPy
test double (tests/numpy_engine.py); what is under test is the product's own host logic:
RowShard bookkeeping with features that straddle ranks, the Chan merge of per-rank feature
statistics, the Gram all-reduce, the per-step pivot candidate all-gather with global
indices, the Theta/cnt all-reduce and the field all-gather -- against the single-process
run and the reference's golden fixture."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _cuts(n, world, uneven):
    """first rows of the ranks' blocks (+ n at the end): equal blocks, or seeded unequal ones that cut through features"""
    if not uneven:
        return [r * (n // world) for r in range(world)] + [n]
    rng = np.random.default_rng(n + world)
    inner = np.sort(rng.choice(np.arange(1, n), size=world - 1, replace=False))
    return [0] + inner.tolist() + [n]


def _worker(rank, world, port, fixture, out_dir, bcast=False, candidates=False, uneven=False):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.conftest import load_golden
        from tests.numpy_engine import CandidateEngine, NumpyEngine
        g = load_golden(fixture)
        X = g['X']
        n = X.shape[0]
        cuts = _cuts(n, world, uneven)
        row0, n_loc = cuts[rank], cuts[rank + 1] - cuts[rank]
        assert uneven or n_loc * world == n
        spr = SPR(np.ascontiguousarray(X[row0:row0 + n_loc]), g['n_features'], None,
                  shard=RowShard(row0, n, broadcast_basis=bcast), engine=CandidateEngine() if candidates else NumpyEngine())
        spr.fit(scale_type=g['scale_type'], axis_cnt=g['axis_cnt'], select_modes=g['select_modes'],
                n_modes=g['n_modes'])
        mask = g.get('mask')
        C = spr.optimal_placement(mask=None if mask is None else mask[row0:row0 + n_loc])
        spr.train(C)
        A3, S3 = spr.predict(list(g['ys']))
        X3 = spr.reconstruct(A3)
        pending = spr.reconstruct(A3, to_host=False, wait=False)      # gather left in flight, joined by wait()
        assert pending.shape == (3, n)
        np.testing.assert_array_equal(pending.wait().numpy().T, X3)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), piv=spr.sensors_, Sigma=spr.Sigma_r, X3=X3, A3=A3,
                 S3=S3, Theta=spr.Theta, X_cnt=spr.X_cnt, X_scl=spr.X_scl, Ur=spr.Ur, Ar=spr.Ar, C_shape=C.shape,
                 gap=spr.pivot_gap_)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('fixture,world', [('g2_num4', 2), ('g2_num4', 3), ('g3_num8', 2), ('g2_num4_mask', 2), ('g4_num5', 3),
                                           ('g5_range', 2), ('g5_l2norm', 3), ('g5_median', 3), ('g6_axisnone', 2),
                                           ('cond_1e7', 2), ('cond_1e5', 3)])   # second-stage Gram all-reduce
def test_sharded_path_matches_reference(tmp_path, fixture, world):
    _run_sharded(tmp_path, fixture, world, False)


@pytest.mark.parametrize('uneven,candidates', [(False, False), (True, False), (False, True), (True, True)])
def test_sharded_path_eight_ranks_nine_features(tmp_path, uneven, candidates):
    """round 6 (VERDICT r05 #1c): world = 8 -- BASELINE configs 4 and 5 run on the 8 GPUs of one node -- on the 9-feature fixture
    g7_f9_num6 (9 x 1000 rows): with eight equal blocks of 1125 rows EVERY interior block boundary falls inside a feature, with
    the seeded unequal cuts blocks also span more than one feature; plain driver and candidate-set model of the placement."""
    fixture = 'g7_f9_num6'
    if not uneven:
        from tests.conftest import load_golden
        n, F = load_golden(fixture)['X'].shape[0], 9
        cuts = _cuts(n, 8, False)
        assert all(c % (n // F) for c in cuts[1:-1])                        # no block boundary on a feature boundary
    _run_sharded(tmp_path, fixture, 8, False, candidates=candidates, uneven=uneven)


def test_sharded_path_with_basis_broadcast(tmp_path):       # RowShard(broadcast_basis=True): rank 0's eigen-solve wins
    _run_sharded(tmp_path, 'g3_num8', 2, True)


@pytest.mark.parametrize('fixture,world', [('g3_num8', 2), ('g3_num8', 4), ('g2_num4_mask', 2)])
def test_sharded_path_on_the_candidate_model(tmp_path, fixture, world):
    """the same fixtures with the candidate-set model under the SPR class: optimal_placement runs the epoch-sweep driver
    (SPR.placement_pools) over gloo, batches of 4 steps, 16-row blocks -- the reference's sensors"""
    _run_sharded(tmp_path, fixture, world, False, candidates=True)


@pytest.mark.parametrize('fixture,world', [('g2_num4', 2), ('g2_num4', 3), ('g3_num8', 3), ('g1_num4', 2), ('g4_num5', 4),
                                           ('g6_axisnone', 3), ('g5_median', 2), ('cond_1e5', 2)])
def test_sharded_path_with_unequal_blocks(tmp_path, fixture, world):
    """row blocks of different sizes (seeded cuts anywhere, also inside a feature; g4's 999 rows do not divide by 4): same
    sensors, basis and fields; the field all-gather pads to the largest block and packs afterwards, the blocks' layout
    rides on fit()'s one all-reduce (no extra collective: test_one_collective_per_fit_and_per_reconstruct)"""
    _run_sharded(tmp_path, fixture, world, False, uneven=True)


@pytest.mark.parametrize('fixture,world', [('g3_num8', 3), ('g2_num4_mask', 2)])
def test_unequal_blocks_on_the_candidate_model(tmp_path, fixture, world):
    """unequal row blocks under the candidate-set model of the placement (every rank keeps its own pool over ITS rows)"""
    _run_sharded(tmp_path, fixture, world, False, candidates=True, uneven=True)


def _gap_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.conftest import load_golden
        from tests.numpy_engine import NumpyEngine
        g = load_golden('g1_num4')
        X = g['X']
        n = X.shape[0]
        row0, n_loc = (0, 10) if rank == 0 else (11, 9)              # row 10 belongs to nobody
        spr = SPR(np.ascontiguousarray(X[row0:row0 + n_loc]), g['n_features'], None, shard=RowShard(row0, n),
                  engine=NumpyEngine())
        try:
            spr.fit(select_modes='number', n_modes=3)              # round 5: the first fit() already refuses the layout
            msg = ''
        except ValueError as e:
            msg = str(e)
        with open(os.path.join(out_dir, f'gap{rank}.txt'), 'w') as fh:
            fh.write(msg)
    finally:
        dist.destroy_process_group()


def test_blocks_that_do_not_cover_the_rows_are_refused_on_every_rank(tmp_path):
    """a hole between two ranks' blocks: the first fit() raises the same ValueError on ALL ranks (the table of blocks comes out
    of its all-reduce, identical everywhere) -- before placement, train or predict can use a wrong global index, and nobody
    is left waiting in a collective"""
    mp.spawn(_gap_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = [open(tmp_path / f'gap{r}.txt').read() for r in range(2)]
    assert msgs[0] == msgs[1] and 'do not cover' in msgs[0] and '[[0, 10], [11, 9]]' in msgs[0]


def _run_sharded(tmp_path, fixture, world, bcast, candidates=False, uneven=False):
    from tests.conftest import load_golden
    from tests.parity import REL_FRO, align_signs, rel_fro
    g = load_golden(fixture)
    n = g['X'].shape[0]
    if n % world and not uneven:
        pytest.skip('rows do not divide')
    mp.spawn(_worker, args=(world, _free_port(), fixture, str(tmp_path), bcast, candidates, uneven), nprocs=world, join=True)
    outs = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
    cuts = _cuts(n, world, uneven)
    for r, o in enumerate(outs):
        # replicated results identical on every rank
        np.testing.assert_array_equal(o['piv'], outs[0]['piv'])
        np.testing.assert_array_equal(o['X3'], outs[0]['X3'])
        np.testing.assert_array_equal(o['Theta'], outs[0]['Theta'])
        # sharded attributes are the local rows
        sl = slice(cuts[r], cuts[r + 1])
        np.testing.assert_allclose(o['X_cnt'], g['X_cnt'][sl], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(o['X_scl'], g['X_scl'][sl], rtol=1e-12)
        sg = align_signs(o['Ar'], g['Ar'])
        np.testing.assert_allclose(o['Ur'] * sg, g['Ur_after_placement' if 'mask' in g else 'Ur'][sl], atol=1e-8)
    o = outs[0]
    assert tuple(o['C_shape']) == tuple(g['C_shape'])
    np.testing.assert_array_equal(o['piv'], g['piv'])                       # global indices, exact, ordered
    np.testing.assert_allclose(o['Sigma'], g['Sigma_r'], rtol=1e-9)
    assert o['X3'].shape == (n, 3)
    assert rel_fro(o['X3'], g['X_rec3']) <= REL_FRO
    sg = align_signs(o['Ar'], g['Ar'])
    np.testing.assert_allclose(o['A3'] * sg, g['Ar_pred3'], atol=1e-7 * np.abs(g['Ar_pred3']).max())
    np.testing.assert_allclose(o['S3'], g['Ar_sigma3'], rtol=1e-6, atol=1e-9 * np.abs(g['Ar_sigma3']).max())
    assert o['gap'].min() > 1e-9


def _gem_worker(rank, world, port, fixture, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.conftest import load_golden_gem
        from tests.numpy_engine import NumpyEngine
        g = load_golden_gem(fixture)
        n = g['X'].shape[0]
        cuts = _cuts(n, world, n % world != 0)                           # 900 rows over 8 ranks: seeded unequal blocks
        row0, n_loc = cuts[rank], cuts[rank + 1] - cuts[rank]
        sl = slice(row0, row0 + n_loc)
        spr = SPR(np.ascontiguousarray(g['X'][sl]), g['n_features'], g['xyz'], shard=RowShard(row0, n),
                  engine=NumpyEngine())
        spr.fit(basis=(np.ascontiguousarray(g['Ur'][sl]), g['Ar']))      # the reference's own basis signs (see parity.py)
        mask = g.get('mask')
        C = spr.optimal_placement(calc_type='gem', n_sensors=g['n_sensors'], d_min=g['d_min'],
                                  mask=None if mask is None else mask[sl])
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), piv=spr.sensors_, C_shape=C.shape)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('fixture,world', [('gem_dmin', 2), ('gem_mask', 3), ('gem_xz_full', 2), ('gem_dmin', 8), ('gem_mask', 8)])
def test_sharded_gem_matches_reference(tmp_path, fixture, world):
    """GEM placement over row shards: the d_min exclusion and the search mask act on every rank's own rows, the
    pick records travel through the same all-gather as the QR placement (features straddle the shard cuts)."""
    from tests.conftest import load_golden_gem
    g = load_golden_gem(fixture)
    assert g['X'].shape[0] % world == 0 or world == 8
    mp.spawn(_gem_worker, args=(world, _free_port(), fixture, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        o = np.load(tmp_path / f'rank{r}.npz')
        np.testing.assert_array_equal(o['piv'], g['gem_piv'])
        assert tuple(o['C_shape']) == tuple(g['C_shape'])


def _gem_ridge_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.numpy_engine import NumpyEngine
        Ur, xyz, F, r = _gem_ridge_case()
        n = Ur.shape[0]
        n_loc = n // world
        row0 = rank * n_loc
        sl = slice(row0, row0 + n_loc)
        spr = SPR(np.zeros((n_loc, 4)), F, xyz, shard=RowShard(row0, n), engine=NumpyEngine())
        spr.fit(basis=(np.ascontiguousarray(Ur[sl]), np.eye(4, r)))
        spr.optimal_placement(calc_type='gem', n_sensors=r + 4, d_min=0.05)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), piv=spr.sensors_)
    finally:
        dist.destroy_process_group()


def _gem_ridge_case():
    rng = np.random.default_rng(77)
    n_points, F, r = 120, 3, 5
    n = n_points * F
    Ur, _ = np.linalg.qr(rng.standard_normal((n, r)) * (1.0 + 3.0 * rng.random((n, 1))))
    return Ur, rng.random((n_points, 3)), F, r


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_gem_beyond_rank(tmp_path, world):
    """GEM with more sensors than r-1 over row shards: the picked rows are collected with an all-reduce, every extra
    pick is agreed on from the all-gathered per-rank records; same sensors as the oracle's ridge rule."""
    from oracle import spr_oracle as orc
    Ur, xyz, F, r = _gem_ridge_case()
    want, lead = orc.gem_pivots(Ur, r + 4, xyz, F, None, 0.05, ridge=1e-5)
    assert lead.min() > 1e-6
    mp.spawn(_gem_ridge_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for k in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f'rank{k}.npz')['piv'], want)


def _f32_worker(rank, world, port, fixture, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix, RowShard
        from tests.conftest import load_golden
        from tests.numpy_engine import NumpyEngine
        import torch
        g = load_golden(fixture)
        X32 = g['X'].astype(np.float32)
        n = X32.shape[0]
        n_loc = n // world
        row0 = rank * n_loc
        blk = torch.from_numpy(np.ascontiguousarray(X32[row0:row0 + n_loc]))
        spr = SPR(DeviceMatrix(blk, basis='f32'), g['n_features'], None, shard=RowShard(row0, n), engine=NumpyEngine())
        spr.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])
        assert spr.Ur.dtype == np.float32
        spr.train(spr.optimal_placement())
        A3, _ = spr.predict(list(g['ys']))
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), piv=spr.sensors_, X3=spr.reconstruct(A3), Ur=spr.Ur)
    finally:
        dist.destroy_process_group()


def test_sharded_f32_storage(tmp_path):
    """float32 shards (config 5's storage) over two ranks: same sensors and fields as one process on the whole
    float32 matrix -- the collectives only ever carry f64 statistics, Gram blocks and records."""
    from openmeasure_amd.sparse_sensing import SPR
    from tests.conftest import load_golden
    from tests.numpy_engine import NumpyEngine
    from tests.parity import REL_FRO, rel_fro
    fixture, world = 'g3_num8', 2
    g = load_golden(fixture)
    mp.spawn(_f32_worker, args=(world, _free_port(), fixture, str(tmp_path)), nprocs=world, join=True)
    import torch
    from openmeasure_amd.sparse_sensing import DeviceMatrix
    one = SPR(DeviceMatrix(torch.from_numpy(g['X'].astype(np.float32)), basis='f32'), g['n_features'], None,
              engine=NumpyEngine())
    one.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])
    one.train(one.optimal_placement())
    A3, _ = one.predict(list(g['ys']))
    X3 = one.reconstruct(A3)
    outs = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
    n_loc = g['X'].shape[0] // world
    for r, o in enumerate(outs):
        np.testing.assert_array_equal(o['piv'], one.sensors_)
        assert rel_fro(o['X3'], X3) <= REL_FRO and rel_fro(o['X3'], g['X_rec3']) <= 1e-5
        sgn = np.sign(np.sum(o['Ur'].astype(np.float64) * one.Ur[r * n_loc:(r + 1) * n_loc], axis=0))
        np.testing.assert_allclose(o['Ur'] * sgn, one.Ur[r * n_loc:(r + 1) * n_loc], atol=2e-7)


def _loop_worker(rank, world, port, fixture, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.conftest import load_golden
        from tests.numpy_engine import NumpyEngine
        g = load_golden(fixture)
        n = g['X'].shape[0]
        n_loc = n // world
        row0 = rank * n_loc
        spr = SPR(np.ascontiguousarray(g['X'][row0:row0 + n_loc]), g['n_features'], None, shard=RowShard(row0, n),
                  engine=NumpyEngine())
        fields, prev = [], None
        for it in range(3):                                   # the step loop of bench.py: the gather of step k is joined
            spr.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])   # only after the fit of step k+1
            if prev is not None:
                fields.append(prev.wait().numpy().copy())
            prev = spr.reconstruct(spr.Ar[:2] * (it + 1), to_host=False, wait=False)
        fields.append(prev.wait().numpy().copy())
        ref = spr.reconstruct(spr.Ar[:2] * 3)                 # synchronous path, same coefficients as the last step
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), f0=fields[0], f1=fields[1], f2=fields[2], ref=ref)
    finally:
        dist.destroy_process_group()


def test_pipelined_field_gather_loop(tmp_path):
    """bench.py's step loop over two ranks: at most one field all-gather in flight, joined after the next fit."""
    from tests.conftest import load_golden
    g = load_golden('g2_num4')
    mp.spawn(_loop_worker, args=(2, _free_port(), 'g2_num4', str(tmp_path)), nprocs=2, join=True)
    outs = [np.load(tmp_path / f'rank{r}.npz') for r in range(2)]
    for o in outs:
        np.testing.assert_array_equal(o['f2'].T, o['ref'])
        np.testing.assert_allclose(o['f1'] - o['f0'], o['f2'] - o['f1'], rtol=0, atol=1e-9 * np.abs(o['f2']).max())
        np.testing.assert_array_equal(o['f2'], outs[0]['f2'])
    assert outs[0]['f2'].shape == (2, g['X'].shape[0])


def _count_worker(rank, world, port, fixture, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.conftest import load_golden
        from tests.numpy_engine import NumpyEngine
        g = load_golden(fixture)
        X = g['X']
        n = X.shape[0]
        n_loc = n // world
        row0 = rank * n_loc
        calls = []
        real = {k: getattr(dist, k) for k in ('all_reduce', 'all_gather_into_tensor', 'all_gather', 'broadcast')}
        for k, fn in real.items():
            setattr(dist, k, (lambda name, f: (lambda *a, **kw: (calls.append(name), f(*a, **kw))[1]))(k, fn))
        spr = SPR(np.ascontiguousarray(X[row0:row0 + n_loc]), g['n_features'], None, shard=RowShard(row0, n),
                  engine=NumpyEngine())
        spr.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])
        fit0_calls = list(calls)
        del calls[:]
        spr.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])
        fit_calls = list(calls)
        del calls[:]
        X3 = spr.reconstruct(spr.Ar[:3])
        rec_calls = list(calls)
        del calls[:]
        x1 = spr.reconstruct(spr.Ar[0], to_host=False, wait=False).wait()
        rec1_calls = list(calls)
        for k, fn in real.items():
            setattr(dist, k, fn)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), fit=np.array(fit_calls), fit0=np.array(fit0_calls), rec=np.array(rec_calls),
                 rec1=np.array(rec1_calls), X3=X3, x1=x1.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_one_collective_per_fit_and_per_reconstruct(tmp_path, world):
    """north_star: 'a single RCCL all-reduce over xGMI for the Gram matrix and a final all-gather for the reconstructed
    field' -- fit() issues exactly ONE collective (the per-rank statistics ride in rank-indexed slots of the Gram
    buffer), reconstruct() exactly ONE all-gather whatever the number of coefficient vectors (the NumPy engine double has
    no p2p exchange: RowShard(gather='auto') resolves to the collective)."""
    from tests.conftest import load_golden
    fixture = 'g3_num8'
    g = load_golden(fixture)
    mp.spawn(_count_worker, args=(world, _free_port(), fixture, str(tmp_path)), nprocs=world, join=True)
    ref = (g['Ur'] @ g['Ar'][:3].T) * g['X_scl'] + g['X_cnt']
    for r in range(world):
        o = np.load(tmp_path / f'rank{r}.npz')
        assert o['fit'].tolist() == ['all_reduce'], o['fit']
        # the object's FIRST fit also all-gathers a 24-byte digest of every rank's host factors (do the ranks' eigen-solves
        # agree? -- ROM._factors_agree), once
        assert o['fit0'].tolist() == ['all_reduce', 'all_gather_into_tensor'], o['fit0']
        assert o['rec'].tolist() == ['all_gather_into_tensor'] and o['rec1'].tolist() == ['all_gather_into_tensor']
        assert np.linalg.norm(o['X3'] - ref) <= 1e-6 * np.linalg.norm(ref)
        np.testing.assert_allclose(o['x1'][0], o['X3'][:, 0], rtol=1e-12, atol=1e-12)


def _wide_worker(rank, world, port, out_dir, bcast, perturb):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import openmeasure_amd._eigen as ss
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.numpy_engine import NumpyEngine
        rng = np.random.default_rng(5)
        n_points, F, m, r = 300, 3, 128, 16                     # m >= 96: the top-r eigen route (dsytrd + dsterf + r vectors)
        n = n_points * F
        X = rng.standard_normal((n, 40)) @ ((0.8 ** np.arange(40))[:, None] * rng.standard_normal((40, m)))
        X += 1e-3 * rng.standard_normal((n, m))
        n_loc = n // world
        row0 = rank * n_loc
        if perturb and rank == world - 1:
            real = ss._eigvecs_top
            if perturb == 'ulp':                                # another LAPACK: the same vectors up to the last bit
                def other(fac, lam, r_):
                    V = real(fac, lam, r_)
                    V[0, 0] = np.nextafter(V[0, 0], np.inf)
                    return V
            else:                                               # 'route': this host's inverse iterations "fail" -> dsyevd, m columns
                def other(fac, lam, r_):
                    return None
            ss._eigvecs_top = other
        spr = SPR(np.ascontiguousarray(X[row0:row0 + n_loc]), F, None, shard=RowShard(row0, n, broadcast_basis=bcast),
                  engine=NumpyEngine())
        fields = []
        for _ in range(2):                                      # the verdict of the first fit holds for the second
            spr.fit(select_modes='number', n_modes=r)
            spr.optimal_placement()
            fields.append(spr.reconstruct(spr.Ar[:2]))
        np.savez(os.path.join(out_dir, f'wide{rank}.npz'), X=X, piv=spr.sensors_, Vr=spr.Vr, S=spr.Sigma_r, f0=fields[0],
                 f1=fields[1], bc=bool(spr.basis_broadcast_), Ar=spr.Ar)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,bcast,perturb', [(2, True, None), (3, True, 'route'), (2, False, None), (2, False, 'ulp'),
                                                 (3, False, 'route')])
def test_host_factors_agree_or_rank0_wins(tmp_path, world, bcast, perturb):
    """Every rank eigen-solves the all-reduced Gram matrix on its own host.  (i) RowShard(broadcast_basis=True) at m >= 96
    (ADVICE r04): rank 0 alone decides the route and r, one fixed-size broadcast -- also when another rank's route would
    have differed; (ii) without the option, the first fit() all-gathers a digest of the factors and switches to rank 0's
    when a rank differs by one ulp or by its route (round 5) -- silently diverging bases are not possible."""
    from oracle import spr_oracle as orc
    mp.spawn(_wide_worker, args=(world, _free_port(), str(tmp_path), bcast, perturb), nprocs=world, join=True)
    outs = [np.load(tmp_path / f'wide{r}.npz') for r in range(world)]
    X = outs[0]['X']
    st = orc.fit(X, 3, select_modes='number', n_modes=16)
    piv, _ = orc.qr_pivots(st['Ur'])
    for o in outs:
        assert bool(o['bc']) == (bcast or perturb is not None)
        np.testing.assert_array_equal(o['piv'], piv)
        np.testing.assert_array_equal(o['Vr'], outs[0]['Vr'])           # the SAME bits on every rank
        np.testing.assert_array_equal(o['f0'], outs[0]['f0'])
        np.testing.assert_array_equal(o['f1'], o['f0'])
        np.testing.assert_allclose(o['S'], st['Sigma_r'], rtol=1e-8)
    sg = np.sign(np.sum(outs[0]['Ar'] * st['Ar'], axis=0))
    ref = (st['Ur'] @ (outs[0]['Ar'][:2] * sg).T) * st['X_scl'] + st['X_cnt']
    assert np.linalg.norm(outs[0]['f0'] - ref) <= 1e-6 * np.linalg.norm(ref)


def _pool_worker(rank, world, port, out_dir, heavy):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import torch
        from openmeasure_amd.sparse_sensing import pivot_loop
        from tests.numpy_engine import CandidateEngine
        rng = np.random.default_rng(11)
        n, r = 1200, 20
        U = rng.standard_normal((n, r))
        U = U * np.exp(1.5 * rng.standard_normal((n, 1))) if heavy else np.linalg.qr(U)[0]
        n_loc = n // world
        row0 = rank * n_loc
        eng = CandidateEngine()
        st = eng.qr_begin(eng.to_device(U[row0:row0 + n_loc]), row0, r)
        calls = []

        def all_gather(t):
            calls.append(tuple(t.shape))
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t.contiguous())
            return torch.stack(out)
        stats = {}
        sweeps = pivot_loop(eng, st, r, all_gather=all_gather, pools=True, stats=stats)
        np.savez(os.path.join(out_dir, f'pool{rank}.npz'), piv=st['piv'].numpy(), U=U, sweeps=sweeps,
                 pool=stats['pool_sweeps'], kinds=np.array([e[0] for e in eng.log]), n_calls=len(calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,heavy', [(2, True), (3, True), (4, False), (2, False)])
def test_sharded_epoch_sweeps_on_the_candidate_model(tmp_path, world, heavy):
    """_pivot_loop_pooled over gloo: every rank keeps its own pool and chooses its own refreshes (their sequences differ
    from rank to rank), the all-gathered tau certifies the steps, and the order is dgeqp3's -- on the candidate-set model of
    the device protocol, whose blocks are small enough that certification fails and pools run dry."""
    from oracle import spr_oracle as orc
    mp.spawn(_pool_worker, args=(world, _free_port(), str(tmp_path), heavy), nprocs=world, join=True)
    outs = [np.load(tmp_path / f'pool{r}.npz') for r in range(world)]
    ref, _ = orc.qr_pivots(outs[0]['U'])
    for o in outs:
        np.testing.assert_array_equal(o['piv'], ref)
        assert int(o['n_calls']) == int(outs[0]['n_calls'])            # the same collectives on every rank
    if heavy:
        assert sum(int(o['pool']) for o in outs) >= 1


def _exchange_worker(rank, world, port, fixture, out_dir, uneven, drop_after, force=None):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.conftest import load_golden
        from tests.numpy_engine import ExchangeDoubleEngine, NumpyEngine
        g = load_golden(fixture)
        X = g['X']
        n = X.shape[0]
        cuts = _cuts(n, world, uneven)
        row0, n_loc = cuts[rank], cuts[rank + 1] - cuts[rank]
        Xl = np.ascontiguousarray(X[row0:row0 + n_loc])
        ref = SPR(Xl, g['n_features'], None, shard=RowShard(row0, n), engine=NumpyEngine())       # no exchange object: the all-gather
        ref.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])
        A3 = g['Ar_pred3'] * np.sign(np.sum(ref.Ar * g['Ar'], axis=0))
        want = ref.reconstruct(A3)
        want1 = ref.reconstruct(A3[:1])                                  # (one vector: BLAS takes another route than for three)
        eng = ExchangeDoubleEngine(faults=dict(drop_after=drop_after) if (drop_after is not None and rank == world - 1) else None)
        spr = SPR(Xl, g['n_features'], None, shard=RowShard(row0, n), engine=eng)
        assert spr.defer_reconstruct
        if force is not None:                                             # the trial's verdict, whatever the clocks say: both branches
            spr._GATHER_TRIAL_MARGIN = 0.0 if force == 'p2p' else 1e9
        spr.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])
        pf = spr.reconstruct(A3, to_host=False, wait=False)              # the object's FIRST sharded reconstruct, deferred:
        assert not pf.launched and '_p2p' not in spr.__dict__             # nothing of the exchange exists yet
        spr.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])     # ... set-up, first exchange and trial run inside this fit
        assert pf.launched
        got = pf.wait().numpy().T.copy()
        again = spr.reconstruct(A3)                                      # ... and the path chosen carries the next one
        one = spr.reconstruct(A3[:1], to_host=False, wait=False)
        spr.fit(select_modes=g['select_modes'], n_modes=g['n_modes'])
        one = one.wait().numpy().copy()
        tr = spr.__dict__.get('gather_trial_', {})
        np.savez(os.path.join(out_dir, f'ex{rank}.npz'), path=str(spr.gather_path_), ok=bool(np.array_equal(got, want)),
                 ok2=bool(np.array_equal(again, want)), ok1=bool(np.array_equal(one[0], want1[:, 0])),
                 trial=np.array([tr.get('p2p_ms', -1.0), tr.get('rccl_ms', -1.0)]), chosen=str(tr.get('chosen')),
                 failed=str(tr.get('failed', '')), has_px='_p2p' in spr.__dict__, fills=int(getattr(eng, 'filler_calls', 0)),
                 pushes=int(eng.exchanges[0].pushes) if eng.exchanges else -1,
                 sigma_same=bool(np.array_equal(spr.Sigma_r, ref.Sigma_r)))
        spr.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,uneven,drop_after,force', [(2, False, None, None), (3, True, None, 'p2p'), (8, False, None, 'rccl'),
                                                           (8, True, None, 'p2p'), (4, False, None, 'rccl'), (2, False, 1, None),
                                                           (4, True, 1, None), (8, False, 1, None), (3, False, 0, None)])
def test_field_exchange_logic_on_the_exchange_double(tmp_path, world, uneven, drop_after, force):
    """round 6: the host logic AROUND the p2p field exchange at any world size on the CPU (tests/numpy_engine.py,
    ExchangeDoubleEngine: the exchange object's interface, its blocks moved by gloo at join time) -- 'auto' sets the exchange
    up inside a DEFERRED first reconstruct, verifies the first exchange block by block, times both paths (the trial) and every
    rank holds the same two numbers and the same verdict; with a rank whose pushes stop after the first exchange
    (drop_after = 1) the trial's p2p legs fail and ALL ranks drop to the all-gather with the reason; with a rank that never
    pushes (0) the first exchange itself fails, same fall-back; the fields equal the all-gather path's bit for bit throughout."""
    fixture = 'g7_f9_num6' if world == 8 else 'g3_num8'
    mp.spawn(_exchange_worker, args=(world, _free_port(), fixture, str(tmp_path), uneven, drop_after, force), nprocs=world, join=True)
    outs = [np.load(tmp_path / f'ex{r}.npz') for r in range(world)]
    for o in outs:
        assert bool(o['ok']) and bool(o['ok2']) and bool(o['ok1']) and bool(o['sigma_same']), (str(o['path']), o['ok'], o['ok2'], o['ok1'])
        assert str(o['path']) == str(outs[0]['path']) and str(o['chosen']) == str(outs[0]['chosen'])
        np.testing.assert_array_equal(o['trial'], outs[0]['trial'])
    o = outs[0]
    if drop_after is None:
        assert 'first-exchange trial' in str(o['path']) and str(o['chosen']) in ('p2p', 'rccl') and o['trial'].min() > 0
        assert str(o['path']).startswith(str(o['chosen'])) and bool(o['has_px']) and int(o['fills']) == 5   # 1 + 2 x 2 trial legs
        assert force is None or str(o['chosen']) == force
    elif drop_after == 1:
        assert str(o['path']).startswith('rccl (p2p failed: ') and 'arrive' in str(o['failed']) and not bool(o['has_px']), str(o['path'])
    else:
        assert str(o['path']).startswith('rccl (p2p failed its first full-size exchange'), str(o['path'])
        assert not bool(o['has_px'])
