"""One rank of a bench.py-style run on the CPU: started by bench.launch_ranks() from tests/test_bench_launcher.py
with the torchrun environment.  Same shard arithmetic (bench.shard_plan) and the product's own sharded host
logic (RowShard, Gram all-reduce, field all-gather) as bench.py's run_rank(), but over the gloo backend with the
NumPy test double of the kernels (tests/numpy_engine.py) -- there is no GPU here.  Not a product path."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_X(n_glob, m, F, seed=7):
    rng = np.random.default_rng(seed)
    k = 6
    X = rng.standard_normal((n_glob, k)) @ ((0.6 ** np.arange(k))[:, None] * rng.standard_normal((k, m)))
    X += 1e-3 * rng.standard_normal((n_glob, m))
    n_points = n_glob // F
    for f in range(F):
        X[f * n_points:(f + 1) * n_points] = (f + 1) * X[f * n_points:(f + 1) * n_points] + 10 * f
    return X


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, required=True)
    ap.add_argument('--cells', type=int, default=120)
    ap.add_argument('--scaling', default='strong')
    ap.add_argument('--fail-rank', type=int, default=-1)
    ap.add_argument('--hang-rank', type=int, default=-1)
    ap.add_argument('--deaf', action='store_true', help='the hanging rank also ignores SIGTERM')
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    import torch.distributed as dist
    import bench
    from openmeasure_amd.sparse_sensing import SPR, RowShard
    from tests.numpy_engine import NumpyEngine

    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    assert int(os.environ['LOCAL_RANK']) == rank
    if rank == args.fail_rank:
        sys.exit(3)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    assert dist.get_world_size() == args.gpus
    if rank == args.hang_rank:
        if args.deaf:
            import signal
            signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(600)
    wl = dict(cells=args.cells, features=3, m=10, s=4, scaling=args.scaling)
    plan = bench.shard_plan(wl, world, rank, 'auto')
    X = make_X(plan['n_glob'], wl['m'], wl['features'])
    Xl = np.ascontiguousarray(X[plan['row0']:plan['row0'] + plan['n_loc']])
    spr = SPR(Xl, wl['features'], None, shard=RowShard(plan['row0'], plan['n_glob']), engine=NumpyEngine())
    spr.comm_timing = {}
    spr.fit(select_modes='number', n_modes=wl['s'])
    field = spr.reconstruct(spr.Ar[0])
    # the two loop forms of bench.py's timed region: gather joined inside the step / left in flight and joined later
    eng = spr._engine()
    a = eng.to_device(spr.Ar[:1].copy())

    def loop(sync):
        spr.comm_timing.clear()
        prev = None
        for _ in range(3):
            spr.fit(select_modes='number', n_modes=wl['s'])
            if prev is not None and hasattr(prev, 'wait'):
                prev.wait()
            prev = spr.reconstruct(a, to_host=False, wait=sync)
        if hasattr(prev, 'wait'):
            prev.wait()
        return {k: float(np.mean([eng.elapsed_ms(e0, e1) for e0, e1 in v])) for k, v in spr.comm_timing.items()}, \
               {k: len(v) for k, v in spr.comm_timing.items()}

    comm_main, n_main = loop(False)
    comm_sync, n_sync = loop(True)
    comm = bench.comm_summary(comm_main, comm_sync, True, False, wl['features'], wl['m'], world, plan['n_loc'])
    if rank == 0:
        if args.out:
            np.save(args.out, field)
        print(json.dumps(dict(n_gpus=dist.get_world_size(), scaling=plan['scaling'], rows_total=plan['n_glob'],
                              rows_per_gpu=plan['n_loc'], n_points=plan['n_points'], comm=comm,
                              brackets_pipelined=n_main, brackets_sync=n_sync)), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
