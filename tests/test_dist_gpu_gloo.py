"""Two ranks, ONE GPU, gloo: the row-sharded path with the REAL kernels (HipEngine) under it.

RCCL refuses two ranks on one device and the build box has one GPU, so the N > 1 RCCL path itself can only be run by
the driver; what can be checked here is everything around the collectives with the real device code: RowShard blocks
whose features straddle the cut, the Gram all-reduce, the candidate all-gather of the placement, the Theta all-reduce and
the field all-gather, on CUDA tensors (gloo stages them through the host) -- against the reference's golden fixtures."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fixture, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from openmeasure_amd.engine import HipEngine
        from openmeasure_amd.sparse_sensing import SPR, RowShard
        from tests.conftest import load_golden
        g = load_golden(fixture)
        X = g['X']
        n = X.shape[0]
        n_loc = n // world
        row0 = rank * n_loc
        spr = SPR(np.ascontiguousarray(X[row0:row0 + n_loc]), g['n_features'], None, shard=RowShard(row0, n),
                  engine=HipEngine('cuda:0'))
        spr.fit(scale_type=g['scale_type'], axis_cnt=g['axis_cnt'], select_modes=g['select_modes'], n_modes=g['n_modes'])
        C = spr.optimal_placement()
        spr.train(C)
        A3, S3 = spr.predict(list(g['ys']))
        X3 = spr.reconstruct(A3)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), piv=spr.sensors_, Sigma=spr.Sigma_r, X3=X3, A3=A3, Ar=spr.Ar,
                 passes=spr.gram_refine_passes_)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('fixture,world', [('g3_num8', 2), ('g2_num4', 3), ('cond_1e7', 2), ('g5_pareto', 2)])
def test_sharded_hip_path_two_ranks_one_gpu(tmp_path, fixture, world):
    import torch.multiprocessing as mp
    from tests.conftest import load_golden
    from tests.parity import REL_FRO, align_signs, rel_fro
    g = load_golden(fixture)
    n = g['X'].shape[0]
    assert n % world == 0
    mp.spawn(_worker, args=(world, _free_port(), fixture, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
    for o in outs:
        np.testing.assert_array_equal(o['piv'], g['piv'])                   # global indices, exact, ordered
        np.testing.assert_array_equal(o['X3'], outs[0]['X3'])
        np.testing.assert_allclose(o['Sigma'], g['Sigma_r'], rtol=1e-8)
        assert rel_fro(o['X3'], g['X_rec3']) <= REL_FRO
        sg = align_signs(o['Ar'], g['Ar'])
        np.testing.assert_allclose(o['A3'] * sg, g['Ar_pred3'], atol=1e-7 * np.abs(g['Ar_pred3']).max())
    if fixture == 'cond_1e7':
        assert int(outs[0]['passes']) >= 1                                  # the second-stage Gram pass ran, sharded
