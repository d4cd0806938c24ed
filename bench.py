#!/usr/bin/env python3
"""SPR fit+reconstruct throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|...]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path over a synthetic snapshot shard that already sits in
HBM:   SPR.fit(select_modes='number', n_modes=s)  +  SPR.reconstruct(a)  (one vector).
value = bytes of the snapshot matrix (all ranks) / max-over-ranks wall time of K steps.
Weak scaling: every rank holds one workload-sized shard, the global matrix has N times the
cells; per step there is one RCCL all-reduce (per-feature Gram) and one all-gather (field).

The JSON line also carries
  roofline      the dominant kernel (the fused stats+Gram pass), timed live with HIP events on
                its launch stream; algorithmic flops/bytes per launch from SURVEY.md 8(d);
  cpu_baseline  the NumPy/LAPACK oracle (oracle/spr_oracle.py) timed on this host's cores on a
                bounded sample of the same workload (rank 0, N=1 only);
  phases, parity  extra evidence (per-kernel ms, GPU-vs-oracle agreement on the sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs (cells, features, snapshots, sensors); c1 is the reference's own
# CPU-sized case, c2/c3 the single-GPU cases, c3 the one the north_star target is quoted on.
WORKLOADS = {
    'c1': dict(cells=18_362, features=9, m=41, s=14, cpu_cells=18_362),
    'c2': dict(cells=1_000_000, features=4, m=64, s=32, cpu_cells=1_000_000),
    'c3': dict(cells=10_000_000, features=9, m=256, s=64, cpu_cells=100_000),
    'c3s': dict(cells=1_000_000, features=9, m=256, s=64, cpu_cells=100_000),   # c3 at 1/10 of the rows
    # config-5 shape (16 features x 512 snapshots, 128 sensors) in f64 at 1M cells/GPU (65.5 GB): the column-split path
    'c5s': dict(cells=1_000_000, features=16, m=512, s=128, cpu_cells=15_000),
    # config 5 as BASELINE.json states it: 50M cells x 16 features x 512 snapshots over 8 GPUs = 6.25M cells (100M
    # rows) per GPU, which only fits in f32 STORAGE (204.8 GB shard + 51.2 GB basis); arithmetic stays f64
    'c5': dict(cells=6_250_000, features=16, m=512, s=128, cpu_cells=15_000, storage='f32'),
    'c5s32': dict(cells=1_000_000, features=16, m=512, s=128, cpu_cells=15_000, storage='f32'),
    'c3s32': dict(cells=1_000_000, features=9, m=256, s=64, cpu_cells=100_000, storage='f32'),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F64_PEAK_TF = 78.6     # AMD public spec for MI355X FP64 matrix; tools/mfma_probe measures 73.6 on the box


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default=os.environ.get('SPR_BENCH_WORKLOAD', 'c3'), choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline / parity leg')
    ap.add_argument('--extra', action='store_true', help='also time placement/train/predict')
    ap.add_argument('--sync-gather', action='store_true',
                    help='join the field all-gather at the end of every step instead of overlapping it with the next Gram pass')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from openmeasure_amd.engine import HipEngine
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix, RowShard
    from openmeasure_amd.synth import make_R

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        log(f'warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE')
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get('SPR_FORCE_DIST', '0') == '1'      # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local_rank}'))

    wl = WORKLOADS[args.workload]
    F, m, s = wl['features'], wl['m'], wl['s']
    f32 = wl.get('storage') == 'f32'
    B = 4 if f32 else 8                           # bytes per stored element of X and Ur
    cells_loc = wl['cells']                       # cells added per rank (weak scaling)
    n_points = cells_loc * world                  # global cells
    n_glob = n_points * F
    n_loc = cells_loc * F                         # rows per rank: contiguous block of the global matrix
    row0 = rank * n_loc
    seed, eps = 1234, 1e-3

    eng = HipEngine(f'cuda:{local_rank}')
    R = eng.to_device(make_R(m, s, seed=seed))
    t0 = time.time()
    Xd = eng.synth(n_loc, m, row0, n_points, R, eps, seed, dtype=torch.float32 if f32 else None)
    torch.cuda.synchronize()
    log(f'[rank {rank}] generated {n_loc} x {m} {"f32" if f32 else "f64"} shard ({n_loc * m * B / 1e9:.2f} GB) in {time.time() - t0:.2f}s')

    shard = RowShard(row0, n_glob, force_collectives=force_dist) if (world > 1 or force_dist) else None
    spr = SPR(DeviceMatrix(Xd), F, None, shard=shard, engine=eng)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    spr.fit(select_modes='number', n_modes=s)     # first call: allocations, RCCL warm-up
    a_d = eng.to_device(spr.Ar[:1].copy())        # (1, r) coefficient vector, resident

    def done(f):
        return f.wait() if hasattr(f, 'wait') else f

    def step(prev=None, timers=None):
        if timers is not None:
            timers.append((eng.time_next('stats_gram'), eng.time_next('project'), eng.time_next('reconstruct')))
        spr.fit(select_modes='number', n_modes=s)
        done(prev)                                # the previous field's all-gather ran under this fit: join it now
        # field all-gather left in flight: it overlaps the next step's (MFMA-bound) Gram pass
        return spr.reconstruct(a_d, to_host=False, wait=args.sync_gather)

    field = None
    for _ in range(args.warmup):
        field = step(field)
    done(field)
    field = None
    timers = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        field = step(field, timers)
    field = done(field)                           # the last gather joins the compute stream inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    if world > 1 or force_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=eng.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = 1e3 * dt / args.steps
    x_bytes = float(n_glob) * m * B
    value = x_bytes / (dt / args.steps) / 1e9

    k_ms = {k: float(np.mean([tm[i][0].elapsed_time(tm[i][1]) for tm in timers]))
            for i, k in enumerate(('stats_gram', 'project', 'reconstruct'))}
    r = spr.r
    # per-launch algorithmic work of each kernel on THIS rank's shard
    alg = {
        'stats_gram': dict(bytes=n_loc * m * B + n_loc * 8, flops=float(n_loc) * m * m),
        'project': dict(bytes=n_loc * m * B + n_loc * r * B, flops=2.0 * n_loc * m * r),
        'reconstruct': dict(bytes=n_loc * r * B + 2 * n_loc * 8, flops=2.0 * n_loc * r),
    }
    phases = {k: dict(ms=round(k_ms[k], 4), GBs=round(alg[k]['bytes'] / k_ms[k] / 1e6, 1),
                      TFLOPs=round(alg[k]['flops'] / k_ms[k] / 1e9, 3)) for k in k_ms}
    dom = max(k_ms, key=k_ms.get)
    t_bytes = alg[dom]['bytes'] / (HBM_PEAK_GBS * 1e9)
    t_flops = alg[dom]['flops'] / (MFMA_F64_PEAK_TF * 1e12)
    if t_flops > t_bytes:
        roof = dict(kernel=dom, bound='mfma', achieved=round(alg[dom]['flops'] / k_ms[dom] / 1e9, 3),
                    peak=MFMA_F64_PEAK_TF, unit='TFLOP/s')
    else:
        roof = dict(kernel=dom, bound='hbm', achieved=round(alg[dom]['bytes'] / k_ms[dom] / 1e6, 1),
                    peak=HBM_PEAK_GBS, unit='GB/s')
    roof['frac'] = round(roof['achieved'] / roof['peak'], 4)
    roof['traffic'] = None                           # HBM bytes/launch from separate rocprofv3 --pmc passes
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            pm = json.load(f).get(args.workload)
        if pm and dom == 'stats_gram' and world == 1:
            roof['traffic'] = pm['hbm_bytes']
            roof['algorithmic_bytes'] = int(alg[dom]['bytes'])
            roof['traffic_source'] = 'profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE, separate passes'
    except (OSError, ValueError):
        pass
    roof['ms'] = round(k_ms[dom], 4)
    # whole-step algorithmic bytes: (2m + 2r) n B for X and Ur + 8 n (field, f64) + 16 n (row means)
    step_bytes = (2 * m + 2 * r) * float(n_glob) * B + 24.0 * n_glob
    hbm_frac = step_bytes / (dt / args.steps) / (world * HBM_PEAK_GBS * 1e9)

    extra = {}
    if args.extra:
        def timed3(fn):
            """median wall time of three calls, each bracketed by barrier + device sync"""
            ts = []
            for _ in range(3):
                barrier(); t_a = time.perf_counter()
                fn()
                barrier(); ts.append(time.perf_counter() - t_a)
            return 1e3 * sorted(ts)[1]
        # two untimed calls first: code objects, allocator growth and the interpreter's first full GC pass all
        # land in the first two placements (tools/placement_probe.py: 180 / 105 / 55 / 55 / 55 ms at config 3)
        spr.optimal_placement()
        piv_first = spr.sensors_.copy()
        spr.optimal_placement()
        t_place = timed3(spr.optimal_placement)
        assert np.array_equal(piv_first, spr.sensors_), 'pivots not reproducible run to run'
        C = spr._placed[0]
        spr.train(C)
        t_train = timed3(lambda: spr.train(C))
        rows = eng.to_device(spr.sensors_, dtype=torch.int64)
        yv = eng.to_host(eng.synth_gather(rows, n_points, m, R, eps, seed))
        y = np.zeros((s, 3)); y[:, 0] = yv; y[:, 2] = spr.sensors_ // n_points
        spr.predict(y)
        t_pred = timed3(lambda: spr.predict(y))
        extra = dict(optimal_placement_ms=round(t_place, 3), train_ms=round(t_train, 3), predict_ms=round(t_pred, 3),
                     min_pivot_gap=float(spr.pivot_gap_.min()), pivot_sweeps=int(spr.pivot_sweeps_),
                     timing='median of 3 calls after 2 warm-up calls')

    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import spr_oracle as orc
        cc = min(wl['cpu_cells'], cells_loc)
        idx = torch.cat([torch.arange(f * n_points, f * n_points + cc, device=eng.device) for f in range(F)])
        Xs = eng.to_host(Xd[idx])                   # first cc cells of every feature: a valid (cc*F) x m problem
        t1 = time.perf_counter()
        # an f32-stored sample goes to the oracle widened: the reference itself would run in f32 on it
        Xo = Xs.astype(np.float64) if f32 else Xs
        xr_cpu, st_cpu = orc.fit_reconstruct_timed(Xo, F, s)
        t_cpu = time.perf_counter() - t1
        try:
            import threadpoolctl
            cores = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] or [1])
        except Exception:
            cores = os.cpu_count()
        cpu = dict(value=round(Xs.nbytes / t_cpu / 1e9, 4), unit='GB/s', cores=int(cores), kind='port',
                   sample=f'{cc} cells x {F} features x {m} snapshots ({Xs.nbytes / 1e6:.0f} MB), s={s}: '
                          f'oracle fit+reconstruct {t_cpu:.2f} s (same generator, first {cc} cells per feature)')
        # parity on the same sample: GPU path vs oracle
        sp2 = SPR(Xs, F, None, engine=eng)
        sp2.fit(select_modes='number', n_modes=s)
        xr_gpu = sp2.reconstruct(st_cpu['Ar'][0])   # same coefficients need the same sign convention:
        sg = np.sign(np.sum(sp2.Ur * st_cpu['Ur'], axis=0))
        xr_gpu = sp2.reconstruct(st_cpu['Ar'][0] * sg)
        sp2.optimal_placement()
        # f32 storage: the sensors are those of the STORED basis (oracle's dgeqp3 on it, widened to f64)
        piv_cpu, _ = orc.qr_pivots(sp2.Ur.astype(np.float64) if f32 else st_cpu['Ur'])
        parity = dict(sensors_equal=bool(np.array_equal(sp2.sensors_, piv_cpu)),
                      field_rel_fro=float(np.linalg.norm(xr_gpu - xr_cpu) / np.linalg.norm(xr_cpu)),
                      sigma_rel=float(np.max(np.abs(sp2.Sigma_r - st_cpu['Sigma_r']) / st_cpu['Sigma_r'])),
                      sigma1_over_sigmas=float(sp2.Sigma_r[0] / sp2.Sigma_r[-1]))

    if rank == 0:
        out = {
            'metric': 'SPR fit+reconstruct throughput (GB/s snapshot matrix)', 'value': round(value, 2),
            'unit': 'GB/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f"{args.workload}: {cells_loc} cells/GPU x {F} features x {m} snapshots, "
                                   f"{s} modes/sensors, {'f32 storage / f64 arithmetic' if f32 else 'f64'}, rows sharded over {world} GPU(s)",
                       'rows_per_gpu': n_loc, 'snapshot_GB_per_gpu': round(n_loc * m * B / 1e9, 3), 'storage': 'f32' if f32 else 'f64'},
            'hbm_roofline_frac_step': round(hbm_frac, 4),
            'roofline': roof, 'cpu_baseline': cpu, 'phases': phases, 'parity': parity,
        }
        if extra:
            out['extra'] = extra
        out['peak_hbm_GB'] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
