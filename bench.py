#!/usr/bin/env python3
"""SPR fit+reconstruct throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|...]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path over a synthetic snapshot shard that already sits in
HBM:   SPR.fit(select_modes='number', n_modes=s)  +  SPR.reconstruct(a)  (one vector).
value = bytes of the snapshot matrix (all ranks) / max-over-ranks wall time of K steps.

N > 1.  One process per GPU over RCCL.  When the ranks' environment (WORLD_SIZE/RANK) is absent,
`python bench.py --gpus N` starts the N rank processes ITSELF -- before anything in this process
touches the GPU -- waits for them and relays rank 0's JSON line; a failed rank ends the run with a
non-zero exit code.  The default workload for N > 1 is BASELINE config 4: the SAME 10M-cell x 9 x 256
matrix as config 3, row-sharded over the N GPUs (strong scaling: 90M/N rows per rank, features
straddle the shards); `--scaling weak` gives every rank a full workload-sized shard instead.
Per step there is one RCCL all-reduce (per-feature Gram) and one exchange of the field -- the one RowShard(gather='auto') chose
(SDMA pushes into peer-mapped buffers, or RCCL's all-gather: the library times both at its first exchange); the other one is
timed as well when the HBM allows it (comm.paths).  N = 1: ms_per_step / value are the SYNCHRONOUS step (fit, then reconstruct
launched and complete before the next fit); ms_per_step_pipelined / value_pipelined the same K steps with the asynchronous
reconstruct(to_host=False, wait=False), whose launch the library defers into the next fit()'s host gap.

The JSON line also carries
  roofline      the dominant kernel (the fused stats+Gram pass), timed live with HIP events on
                its launch stream; algorithmic flops/bytes per launch from SURVEY.md 8(d);
                roofline_per_rank repeats it for every rank when N > 1;
  cpu_baseline  the NumPy/LAPACK oracle (oracle/spr_oracle.py) timed on this host's cores on a
                sample of the same workload sized for about SPR_BENCH_CPU_BUDGET_S = 30 s from a
                short probe (rank 0, N=1 only);
  hbm           hipMemGetInfo at the end of the run, PyTorch's peaks, the exchange buffers;
  phases, parity  extra evidence (per-kernel ms, GPU-vs-oracle agreement on the sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs (cells, features, snapshots, sensors); c1 is the reference's own
# CPU-sized case, c2/c3 the single-GPU cases, c3 the one the north_star target is quoted on,
# c4 the same matrix sharded over the GPUs of the node, c5 the one that exceeds a single GPU.
WORKLOADS = {
    'c1': dict(cells=18_362, features=9, m=41, s=14, cpu_cells=18_362),
    'c2': dict(cells=1_000_000, features=4, m=64, s=32, cpu_cells=1_000_000),
    'c3': dict(cells=10_000_000, features=9, m=256, s=64, cpu_cells=100_000),
    # config 4: config 3's matrix, 10M cells IN TOTAL, split into 10M/N cells' worth of rows per rank
    'c4': dict(cells=10_000_000, features=9, m=256, s=64, cpu_cells=100_000, scaling='strong'),
    'c3s': dict(cells=1_000_000, features=9, m=256, s=64, cpu_cells=100_000),   # c3 at 1/10 of the rows
    'c4s': dict(cells=1_000_000, features=9, m=256, s=64, cpu_cells=100_000, scaling='strong'),   # c4 at 1/10 (rehearsals)
    # config-3 shape at 1M cells with a designed sigma_1/sigma_s = 1e6: the conditioning refinement of fit() at scale
    # (one more pass over X: projection onto all m first-stage vectors + Gram of the result), reported as refine_ms
    'c3k': dict(cells=1_000_000, features=9, m=256, s=64, cpu_cells=100_000, ratio=1e6, eps=1e-9),
    # config-5 shape (16 features x 512 snapshots, 128 sensors) in f64 at 1M cells/GPU (65.5 GB): the column-split path
    'c5s': dict(cells=1_000_000, features=16, m=512, s=128, cpu_cells=15_000),
    # config 5 as BASELINE.json states it: 50M cells x 16 features x 512 snapshots over 8 GPUs = 6.25M cells (100M
    # rows) per GPU, which only fits in f32 STORAGE (204.8 GB shard + 51.2 GB basis); arithmetic stays f64.
    # Weak by construction: at --gpus 8 it IS config 5, at --gpus 1 it is one GPU's share of it.
    'c5': dict(cells=6_250_000, features=16, m=512, s=128, cpu_cells=15_000, storage='f32'),
    'c5s32': dict(cells=1_000_000, features=16, m=512, s=128, cpu_cells=15_000, storage='f32'),
    'c3s32': dict(cells=1_000_000, features=9, m=256, s=64, cpu_cells=100_000, storage='f32'),
}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F64_PEAK_TF = 78.6     # AMD public spec for MI355X FP64 matrix; tools/mfma_probe measures 73.6 on the box
HBM_MEASURED_GBS = 6290.0   # MI355X_MICROARCH.md chip table: float4 copy, 79 % of spec
MFMA_F64_MEASURED_TF = 73.6 # tools/mfma_probe.hip: 64 cycles per v_mfma_f64_16x16x4_f64 and SIMD at the clock held under load


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------
# shard arithmetic and rank launcher (also driven by tests/test_bench_launcher.py over gloo)
# ---------------------------------------------------------------------------------------------
def shard_plan(wl, world, rank, scaling='auto'):
    """Rows of the global feature-major matrix held by `rank` of `world`.

    strong: the workload's cells are the TOTAL (config 4); every rank holds n_glob/world contiguous rows, so features
            straddle the shard boundaries.  The cell count is rounded down to a multiple of `world` so that all
            ranks hold the same number of rows (the field all-gather needs equal shards).
    weak:   every rank adds one workload-sized block of cells; the global matrix has world x the cells.
    """
    mode = wl.get('scaling', 'weak') if scaling == 'auto' else scaling
    if mode not in ('weak', 'strong'):
        raise ValueError(f'scaling={scaling!r}')
    F = wl['features']
    if mode == 'strong':
        n_points = (wl['cells'] // world) * world
        if n_points < world:
            raise ValueError('fewer cells than ranks')
    else:
        n_points = wl['cells'] * world
    n_glob = n_points * F
    n_loc = n_glob // world
    assert n_loc * world == n_glob
    return dict(scaling=mode, n_points=n_points, n_glob=n_glob, n_loc=n_loc, row0=rank * n_loc)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, env_extra=None, timeout=None, relay=sys.stdout, grace=10.0):
    """Start `n` rank processes running `argv` (a full command line), one per GPU, with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and wait for them.  The caller must not have touched
    the GPU: the children are fresh processes (never an exec of this one).  Rank 0's stdout is relayed line by line;
    the other ranks' stdout goes to stderr.  Returns the exit code: 0 when every rank succeeded, else the first
    failing rank's code; the remaining ranks get SIGTERM by PID and, `grace` seconds later, SIGKILL -- a rank stuck in a
    collective must not hang the launcher."""
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on these hosts (RCCL needs it)
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr,
                                      stderr=sys.stderr, text=(rank == 0)))
    t_end = None if timeout is None else time.time() + timeout
    rc = 0
    out0 = procs[0].stdout
    import threading

    def pump():
        for line in out0:                                      # rank 0's JSON line goes to stdout, anything else to stderr
            dst = relay if line.lstrip().startswith('{') else sys.stderr
            dst.write(line)
            dst.flush()
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    live = set(range(n))
    kill_at = None                                             # after a failure / timeout: SIGTERM now, SIGKILL at this time

    def stop_others(why):
        nonlocal kill_at
        log(f'[launcher] {why}; stopping the other ranks')
        for o in live:
            procs[o].terminate()
        kill_at = time.time() + grace

    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                if kill_at is None:
                    stop_others(f'rank {r} exited with code {code}')
        if t_end is not None and time.time() > t_end and live and kill_at is None:
            rc = rc or 124
            stop_others('timeout')
        if kill_at is not None and time.time() > kill_at and live:
            # a rank stuck inside a collective or the driver does not leave on SIGTERM: kill it, never wait forever
            log(f'[launcher] ranks {sorted(live)} ignored SIGTERM for {grace:.0f} s: SIGKILL')
            for o in live:
                procs[o].kill()
            for o in list(live):
                procs[o].wait()
                live.discard(o)
            rc = rc or 137
        time.sleep(0.05)
    th.join(timeout=5)
    return rc


def _visible_limit(env=None):
    """How many devices the visibility variables leave (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES,
    comma-separated indices or UUIDs; an empty string hides every device).  None when none of them is set."""
    env = os.environ if env is None else env
    lim = None
    for key in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if key in env:
            k = len([t for t in env[key].split(',') if t.strip() != ''])
            lim = k if lim is None else min(lim, k)
    return lim


def count_gpus(env=None):
    """AMD GPUs THIS process's children can use, WITHOUT touching the HIP runtime (the launcher parent must stay GPU-free:
    its children are the only processes that may initialise the device): KFD topology nodes with a gfx target, else the DRM
    render nodes whose PCI vendor is AMD (0x1002), capped by the visibility variables.  None when nothing is readable."""
    import glob
    k = None
    nodes = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    if nodes:
        k = 0
        for p in nodes:
            try:
                with open(p) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                k += int(props.get('gfx_target_version', '0')) != 0
            except (OSError, ValueError):
                pass
    else:
        rd = glob.glob('/sys/class/drm/renderD*')
        if rd:
            k = 0
            for p in rd:
                try:
                    with open(os.path.join(p, 'device', 'vendor')) as f:
                        k += f.read().strip().lower() == '0x1002'
                except OSError:
                    pass
    if k is None:
        return None
    lim = _visible_limit(env)
    return k if lim is None else min(k, lim)


class Watchdog:
    """First-contact armour for multi-GPU runs: a rank that makes no progress for `timeout_s` seconds -- a collective its
    peers never joined, a stream wait on a counter nobody raises -- says where it stands (phase label, step, the collective
    brackets recorded so far, every thread's Python stack) and ends the PROCESS with a non-zero code; the launcher then stops
    the other ranks.  It never re-executes anything.  `beat(label)` is called at every phase boundary of the run."""

    def __init__(self, timeout_s, rank=0, describe=None, exit_code=3):
        import threading
        self.timeout_s, self.rank, self.describe, self.exit_code = float(timeout_s), rank, describe, exit_code
        self.label, self.t, self.on = 'start', time.time(), self.timeout_s > 0
        if self.on:
            threading.Thread(target=self._run, daemon=True, name='spr-watchdog').start()

    def beat(self, label):
        self.label, self.t = label, time.time()

    def stop(self):
        self.on = False

    def _run(self):
        while self.on:
            time.sleep(min(1.0, max(0.05, self.timeout_s / 4)))
            idle = time.time() - self.t
            if self.on and idle > self.timeout_s:
                import faulthandler
                log(f'[rank {self.rank}] WATCHDOG: no progress for {idle:.0f} s in phase {self.label!r}')
                try:
                    if self.describe is not None:
                        log(f'[rank {self.rank}] WATCHDOG: {self.describe()}')
                except Exception as exc:                       # noqa: BLE001 -- the dump must not stop the exit
                    log(f'[rank {self.rank}] WATCHDOG: describe() failed: {exc}')
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                sys.stderr.flush()
                os._exit(self.exit_code)


def comm_summary(comm_main, comm_sync, dist_on, sync_gather, F, m, world, n_loc, gather_path=None, paths=None):
    """The `comm` object of the JSON line from the mean bracket times (ms) of the headline loop and of the sync-gather loop
    (ROM.comm_timing keys 'allreduce', 'gather', 'gather_exposed').  None = no such collective ran."""
    src = comm_sync if comm_sync else comm_main
    comm = dict(allreduce_ms=comm_main.get('allreduce'), gather_ms=src.get('gather'),
                gather_exposed_ms=comm_main.get('gather_exposed', 0.0 if (dist_on and not sync_gather) else None),
                allreduce_bytes=(F * m * m + world * F * 3) * 8 if dist_on else 0,
                gather_bytes_per_rank=n_loc * 8 if dist_on else 0,
                note=('per step over the timed steps; allreduce_ms: mean bracket around the call on the compute stream; '
                      'gather_ms: mean issue to join in the sync-gather loop; gather_exposed_ms: MEDIAN join of the pipelined loop '
                      '(what did not hide under the next Gram pass); gather_exposed_last_ms: the loop\'s final join, which has '
                      'nothing to hide under') if dist_on else 'no collectives at N = 1')
    if dist_on:
        comm['gather_exposed_last_ms'] = (round(comm_main['gather_exposed_last'], 4)
                                          if comm_main.get('gather_exposed_last') is not None else None)
    for k_ in ('allreduce_ms', 'gather_ms', 'gather_exposed_ms'):
        if comm[k_] is not None:
            comm[k_] = round(comm[k_], 4)
    if dist_on:
        # which exchange carried the field in the headline loop ('p2p ...': SDMA pushes into peer-mapped buffers, no compute
        # units; 'rccl ...': the all-gather kernel -- with the reason when p2p was not available), and both timed side by side
        comm['gather_path'] = gather_path
        comm['paths'] = paths
    return comm


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default=os.environ.get('SPR_BENCH_WORKLOAD'), choices=sorted(WORKLOADS),
                    help='default: c3 on one GPU, c4 (config 3 sharded, strong scaling) on several')
    ap.add_argument('--scaling', default='auto', choices=('auto', 'weak', 'strong'),
                    help="auto = the workload's own mode (c4: strong, the others: weak)")
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline / parity leg')
    ap.add_argument('--extra', action='store_true',
                    help='more lines: A/B of the row norms fit() leaves for the placement, reconstruct() with its '
                         "reference output contract (host ndarray), upload rate of a host ndarray X")
    ap.add_argument('--placement-norms', choices=('auto', 'on', 'off'), default='auto',
                    help="developer aid (A/B): SPR.placement_norms -- whether fit()'s projection also leaves the squared row "
                         "norms optimal_placement starts from (default: the class default, 'auto')")
    ap.add_argument('--sync-gather', action='store_true',
                    help='the headline loop joins the field all-gather at the end of every step (default: it overlaps the '
                         'next Gram pass; both forms are timed in every N > 1 run)')
    ap.add_argument('--share-of', type=int, default=0, metavar='N',
                    help="developer aid: run ONE rank's shard of an N-rank job on this GPU (collectives in a 1-rank RCCL "
                         'group); the line says so and is not an N-GPU number')
    ap.add_argument('--share-rank', type=int, default=0, help='which rank of --share-of')
    ap.add_argument('--gather', default=os.environ.get('SPR_BENCH_GATHER', 'auto'), choices=('auto', 'p2p', 'rccl'),
                    help="field exchange of the headline loop (RowShard.gather); with 'auto' the other path is timed as well")
    ap.add_argument('--headline-gather', default='library', choices=('library', 'faster', 'other'),
                    help="N > 1 with --gather auto, where both exchanges are timed: which one the line's value comes from -- the "
                         "library's own choice (default: RowShard(gather='auto') times both at its first exchange and keeps the "
                         "faster, ROM._gather_trial -- what a user of the library gets), the faster of the two as timed by this run, "
                         "or the other one (to exercise that branch)")
    ap.add_argument('--p2p-loopback', type=int, default=0, metavar='L',
                    help='developer aid, with --share-of: the p2p field exchange with L imaginary peers inside this GPU (the '
                         "rank issues the L pushes of its block an (L+1)-rank exchange would, both ends in its own HBM); the "
                         'line says so')
    ap.add_argument('--defer-reconstruct', action='store_true',
                    help="(the library's default since round 6; kept so that older command lines still run) ROM.defer_reconstruct: "
                         "in the pipelined loop the step's reconstruct kernel -- and, sharded, the push of its block -- is enqueued "
                         "in the host gap of the NEXT step's fit() instead of behind this step's projection; every step still runs "
                         'inside the timed region (the last one is flushed before the closing barrier)')
    ap.add_argument('--no-defer-reconstruct', action='store_true',
                    help='switch ROM.defer_reconstruct off (A/B): every reconstruct launches behind its own projection')
    ap.add_argument('--ballast-gb', type=float, default=0.0, metavar='G',
                    help="developer aid (N = 8 rehearsal on one GPU): allocate G GB of HBM before the first fit() and hold them to "
                         "the end -- what an N-rank run adds to one rank's footprint (the persistent copy of the gathered field, "
                         "RCCL's staging and buffers) on top of what --share-of / --p2p-loopback already allocate; the line says so")
    ap.add_argument('--gap-filler', action='store_true',
                    help='switch ROM.gap_filler on (opt-in: fit() re-queues its Gram kernel on part of X into the host gap to '
                         'hold the clock; the line then says so and counts the rows)')
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.workload is None:
        args.workload = 'c3' if args.gpus == 1 else 'c4'
    return args


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  This process never touches the GPU or imports torch:
        # the device count comes from sysfs
        have = count_gpus()
        if have is not None and have < args.gpus and os.environ.get('SPR_BENCH_ONE_GPU') != '1':
            log(f'bench.py: --gpus {args.gpus} but this node shows {have} GPU(s)')
            sys.exit(2)
        sys.exit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    run_rank(args)


def run_rank(args):
    # exactly ONE line may reach stdout (the JSON line of rank 0): RCCL prints a version banner to the C-level stdout
    # when its communicator is created, so file descriptor 1 points at stderr for the whole run and the JSON line is
    # written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.gpus > 1 or args.share_of > 1:
        # Sharded runs use a compute stream, three copy streams (p2p field exchange), RCCL's own stream and a host-transfer
        # stream; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one are
        # served in submission order -- a copy stream's wait for its SDMA engine then sits in front of a kernel of the compute
        # stream.  Read by the runtime when it initialises, i.e. below (measured: profiles/r05_p2p_gap_experiments.txt).
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    import torch
    import torch.distributed as dist
    from openmeasure_amd.engine import HipEngine
    from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix, RowShard
    from openmeasure_amd.synth import make_R

    env_world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if env_world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU '
                         '(or drop WORLD_SIZE and let bench.py start the ranks)')
    # Rehearsal of the N-rank code path on a box with ONE GPU (RCCL refuses two ranks on one device):
    # SPR_BENCH_ONE_GPU=1 puts every rank on cuda:0 and SPR_BENCH_BACKEND=gloo moves the collectives to gloo (staged
    # through the host).  The line says so; such a run measures nothing, it only exercises the code.
    backend = os.environ.get('SPR_BENCH_BACKEND', 'nccl')
    if os.environ.get('SPR_BENCH_ONE_GPU') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    share = args.share_of if args.share_of > 1 else 0
    force_dist = os.environ.get('SPR_FORCE_DIST', '0') == '1' or bool(share)   # exercise the RCCL path with one rank
    if env_world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local_rank}'))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()                 # the RCCL communicator's size is what gets reported
    else:
        world = 1
    assert world == args.gpus, f'RCCL world size {world} != --gpus {args.gpus}'

    wl = WORKLOADS[args.workload]
    F, m, s = wl['features'], wl['m'], wl['s']
    f32 = wl.get('storage') == 'f32'
    B = 4 if f32 else 8                           # bytes per stored element of X and Ur
    if share:
        plan = shard_plan(wl, share, args.share_rank, args.scaling)
    else:
        plan = shard_plan(wl, world, rank, args.scaling)
    n_points, n_glob, n_loc, row0 = plan['n_points'], plan['n_glob'], plan['n_loc'], plan['row0']
    seed, eps = 1234, wl.get('eps', 1e-3)

    if args.p2p_loopback:
        if not share:
            raise SystemExit('bench.py: --p2p-loopback is a diagnostic of --share-of runs')
        os.environ['SPR_P2P_LOOPBACK'] = str(args.p2p_loopback)
        args.gather = 'p2p'
    eng = HipEngine(f'cuda:{local_rank}')
    R = eng.to_device(make_R(m, s, seed=seed, ratio=wl.get('ratio', 1e3)))
    t0 = time.time()
    Xd = eng.synth(n_loc, m, row0, n_points, R, eps, seed, dtype=torch.float32 if f32 else None)
    torch.cuda.synchronize()
    log(f'[rank {rank}] generated rows [{row0}, {row0 + n_loc}) x {m} {"f32" if f32 else "f64"} '
        f'({n_loc * m * B / 1e9:.2f} GB) of {n_glob} global rows in {time.time() - t0:.2f}s')

    if share:
        # one rank's block of an N-rank job, alone: the global row numbering (feature boundaries) of the N-rank job
        shard = RowShard(row0, n_glob, force_collectives=True, partial=True, gather=args.gather)
    else:
        shard = (RowShard(row0, n_glob, force_collectives=force_dist, gather=args.gather)
                 if (world > 1 or force_dist) else None)
    # f32-stored workloads (config 5) also store the basis in f32 -- the explicit storage option; the default would be f64
    spr = SPR(DeviceMatrix(Xd, basis='f32' if f32 else None), F, None, shard=shard, engine=eng)
    if args.placement_norms != 'auto':
        spr.placement_norms = args.placement_norms == 'on'
    if args.gap_filler:
        spr.gap_filler = True
    if args.no_defer_reconstruct:
        spr.defer_reconstruct = False
    elif args.defer_reconstruct:
        spr.defer_reconstruct = True
    ballast = None
    if args.ballast_gb > 0:
        ballast = torch.empty(int(args.ballast_gb * 1e9), dtype=torch.uint8, device=eng.device)
        ballast[::1 << 20].zero_()                          # touched: really resident
        log(f'[rank {rank}] ballast: {ballast.numel() / 1e9:.2f} GB held next to the shard')
    wd = Watchdog(float(os.environ.get('SPR_BENCH_WATCHDOG_S', '300')), rank,
                  lambda: dict(collective_brackets={k: len(v) for k, v in (spr.comm_timing or {}).items()},
                               last_collective=getattr(spr, 'last_comm_', None), gather_path=getattr(spr, 'gather_path_', None)))

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    dist_on = world > 1 or force_dist
    own_stream = dist_on and os.environ.get('SPR_BENCH_STREAM', '1') != '0'
    if own_stream:
        # Sharded runs put the whole step on a stream of their own: on the DEFAULT (null) stream a kernel launched while SDMA
        # pushes of the p2p field exchange are in flight on other streams only starts when they end -- 0.9 ms per step on one
        # rank's block of config 4 (profiles/r05_p2p_gap_experiments.txt: reconstruct -> next Gram pass 0.92 ms on the null
        # stream, 0.05 ms on any other).  SPR_BENCH_STREAM=0 keeps the default stream (A/B).
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(eng.device))
    wd.beat('first fit')
    spr.fit(select_modes='number', n_modes=s)     # first call: allocations, RCCL warm-up
    a_d = eng.to_device(spr.Ar[:1].copy())        # (1, r) coefficient vector, resident
    if dist_on:
        spr.comm_timing = {}                      # event pairs around every collective (ROM.comm_timing)

    def done(f):
        return f.wait() if hasattr(f, 'wait') else f

    fill_rows = []

    def timed_loop(sync_gather):
        """W warm-up steps, then K steps bracketed by barrier + device sync on both sides.  -> (seconds, max over ranks;
        this rank's own seconds; per-step kernel event triples; mean ms of every collective bracket; the last field)"""
        def step(prev=None, timers=None):
            if timers is not None:
                timers.append((eng.time_next('stats_gram'), eng.time_next('project'), eng.time_next('reconstruct')))
            wd.beat('fit')
            spr.fit(select_modes='number', n_modes=s)
            if timers is not None:
                fill_rows.append(int(spr.__dict__.get('_gap_fill_rows', 0)) if spr.gap_filler else 0)
            wd.beat('join of the previous field')
            done(prev)                            # the previous field's exchange ran under this fit: join it now
            wd.beat('reconstruct')
            # default: the exchange is left in flight and overlaps the next step's (MFMA-bound) Gram pass
            return spr.reconstruct(a_d, to_host=False, wait=sync_gather)

        field = None
        for _ in range(args.warmup):
            field = step(field)
        done(field)
        field = None
        timers = []
        if dist_on:
            spr.comm_timing.clear()
        wd.beat('barrier in front of the timed steps')
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            field = step(field, timers)
        wd.beat('last join')
        field = done(field)                       # the last exchange joins the compute stream inside the timed region
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0         # this rank's own time (before the others are waited for)
        wd.beat('barrier behind the timed steps')
        barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            tt = torch.tensor([dt], dtype=torch.float64, device=eng.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        comm = {}
        if dist_on:
            for key, pairs in spr.comm_timing.items():
                ms = [eng.elapsed_ms(e0, e1) for e0, e1 in pairs]
                # the exposed part of a pipelined exchange: the MEDIAN over the joins (the last join of the loop has nothing to
                # hide under -- its exchange is joined right behind its own reconstruct -- and is reported on its own)
                comm[key] = (float(np.median(ms)) if key == 'gather_exposed' else float(np.mean(ms))) if ms else None
                if key == 'gather_exposed' and ms:
                    comm['gather_exposed_last'] = float(ms[-1])
        return dt, dt_own, timers, comm, field

    # N > 1: the field exchange of the headline loop is RowShard(gather=...) as resolved at the first sharded reconstruct()
    # ('auto': p2p when its collective self-test passes AND the library's own first-exchange trial does not find the all-gather
    # faster, ROM._gather_trial); the OTHER path is then timed with the same loops when the HBM left allows it, so one run says
    # what each costs, pipelined and joined inside the step
    path_results = {}

    def hbm_free_all_ranks():
        """bytes a new allocation could still get on the tightest rank: free HBM + what torch's allocator holds cached"""
        free = torch.cuda.mem_get_info(eng.device)[0] + torch.cuda.memory_reserved(eng.device) - torch.cuda.memory_allocated(eng.device)
        tt = torch.tensor([float(free)], dtype=torch.float64, device=eng.device)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        return float(tt.item())

    if dist_on:
        wd.beat('first sharded reconstruct (gather path set-up)')
        done(spr.reconstruct(a_d, to_host=False, wait=True))
        head_path = 'p2p' if spr.gather_path_.startswith('p2p') else 'rccl'
        head_why = spr.gather_path_
        others = []
        if args.gather == 'auto':
            # p2p in the headline: the all-gather is the other one; the all-gather chosen by the library's trial with the p2p buffers
            # still mapped: p2p is
            others = ['rccl'] if head_path == 'p2p' else (['p2p'] if '_p2p' in spr.__dict__ else [])
    else:
        head_path, head_why, others = None, None, []

    # N = 1: the headline is the SYNCHRONOUS form (every step's field complete before the next step starts: comparable with rounds
    # 1-5); the pipelined form -- reconstruct(to_host=False, wait=False), deferred into the next fit()'s host gap, the library's
    # default for that call -- is timed right behind it and reported as ms_per_step_pipelined
    head_sync = args.sync_gather or not dist_on
    dt, dt_own, timers, comm_main, field = timed_loop(head_sync)
    dt_sync, comm_sync = None, {}
    dt_pipe = None
    if dist_on and not args.sync_gather:
        # the same K steps once more with the exchange joined inside every step: what the pipelining hides, and the
        # duration of the exchange itself (its bracket closes behind the join)
        dt_sync, _, _, comm_sync, _ = timed_loop(True)
    elif not dist_on:
        dt_pipe, _, _, _, _ = timed_loop(False)
    if dist_on:
        path_results[head_path] = dict(ms_per_step=round(1e3 * dt / args.steps, 4),
                                       ms_per_step_sync_gather=round(1e3 * (dt_sync if dt_sync is not None else dt) / args.steps, 4),
                                       allreduce_ms=comm_main.get('allreduce'),
                                       gather_ms=(comm_sync or comm_main).get('gather'),
                                       gather_exposed_ms=comm_main.get('gather_exposed'), why=head_why)
        raw_other = None
        for other in others:
            # HBM first (config 5 at N = 8 leaves a few GB): the all-gather wants its staging (world x block) and the block, plus
            # what RCCL allocates for itself; the p2p buffers exist already when p2p is the other one
            need = ((world + 1) * n_loc * 8 + (2 << 30)) if other == 'rccl' else 0
            free = hbm_free_all_ranks()
            if free < need:
                path_results[other] = dict(skipped=f'not timed: the {other} leg needs about {need / 1e9:.1f} GB more HBM per rank '
                                                   f'(staging of the gathered field + RCCL buffers), {free / 1e9:.1f} GB are left on the '
                                                   'tightest rank')
                log(f'[rank {rank}] {other} leg skipped: {path_results[other]["skipped"]}')
                continue
            wd.beat(f'switch to gather path {other}')
            spr.use_gather(other)
            done(spr.reconstruct(a_d, to_host=False, wait=True))
            dt_o, _, _, comm_o, _ = timed_loop(False)
            dt_os, _, _, comm_os, _ = timed_loop(True)
            path_results[other] = dict(ms_per_step=round(1e3 * dt_o / args.steps, 4),
                                       ms_per_step_sync_gather=round(1e3 * dt_os / args.steps, 4),
                                       allreduce_ms=comm_o.get('allreduce'), gather_ms=comm_os.get('gather'),
                                       gather_exposed_ms=comm_o.get('gather_exposed'), why=spr.gather_path_)
            raw_other = (dt_o, comm_o, dt_os, comm_os, spr.gather_path_, other)
        if raw_other is not None:
            # both exchanges have been timed with the same loops (dt is the maximum over the ranks: the same number on every rank).
            # --headline-gather library (default): the line's value is the path the library chose; faster / other: A/B aids
            other = raw_other[5]
            take_other = {'faster': raw_other[0] < 0.99 * dt, 'library': False, 'other': True}[args.headline_gather]
            if take_other:
                head_why = (f'{raw_other[4]} -- chosen over {head_path} by this run (--headline-gather {args.headline_gather}): '
                            f'{1e3 * raw_other[0] / args.steps:.3f} against {1e3 * dt / args.steps:.3f} ms per step')
                wd.beat(f'headline loop once more on the other path ({other})')
                dt, dt_own, timers, comm_main, field = timed_loop(args.sync_gather)     # kernel timers / timeline of THAT path
                dt_sync, comm_sync = raw_other[2], raw_other[3]
                head_path = other
                path_results[other]['ms_per_step'] = round(1e3 * dt / args.steps, 4)
            else:
                wd.beat('back to the headline gather path')
                spr.use_gather(head_path)
                field = done(spr.reconstruct(a_d, to_host=False, wait=True))
        for v in path_results.values():
            for k_ in ('allreduce_ms', 'gather_ms', 'gather_exposed_ms'):
                if v.get(k_) is not None:
                    v[k_] = round(v[k_], 4)
    ms_per_step = 1e3 * dt / args.steps
    # collectives, from event pairs on the stream each one is ordered on (ROM.comm_timing): the ONE all-reduce of fit(), the
    # field exchange of reconstruct() timed where it is joined inside the step, and what of it stays exposed when it is
    # left in flight under the next Gram pass (the wait of the join).  None = no such collective ran (N = 1).
    comm = comm_summary(comm_main, comm_sync, dist_on, args.sync_gather, F, m, world, n_loc, gather_path=head_why,
                        paths=path_results or None)
    px = spr.__dict__.get('_p2p')
    if px is not None:
        px.check()                                    # a join kernel that gave up on a peer: no result line for such a run
    if dist_on and px is not None and px.host_ms['calls']:
        comm['p2p_host_ms_per_gather'] = {k_: round(v / px.host_ms['calls'], 4) for k_, v in px.host_ms.items() if k_ != 'calls'}
        comm['p2p_copy_streams'] = len(px._pool)
        # the first exchange through the mapped buffers was compared block by block with what the peers sent (ROM._p2p_first_exchange)
        comm['p2p_first_exchange'] = px.verified
        comm['p2p_memory'] = px.memory                      # 'coarse' (plain device memory) unless the self-test asked for 'uncached'
    if dist_on:
        # the library's own first-exchange trial of the two paths (ROM._gather_trial): what RowShard(gather='auto') decided from
        comm['gather_trial'] = getattr(spr, 'gather_trial_', None)
    # never print a number for a run that computed garbage: spectrum, basis sample and field must be finite
    fld = field if torch.is_tensor(field) else None
    if not (np.all(np.isfinite(spr.S_[:s])) and bool(torch.isfinite(spr._d['Ur'][:4096].double()).all())
            and (fld is None or bool(torch.isfinite(fld[..., ::max(1, fld.shape[-1] // 65536)]).all()))):
        raise SystemExit('bench.py: non-finite spectrum / basis / field -- no result line')
    n_job = n_loc if share else n_glob            # rows this run actually processed (a share run: one rank's block)
    x_bytes = float(n_job) * m * B
    value = x_bytes / (dt / args.steps) / 1e9

    def ev_ms(a, b):
        try:
            return a.elapsed_time(b)
        except (RuntimeError, ValueError):            # a pair that never got its launch (deferred reconstruct: the last step's)
            return None

    def mean_ms(vals):
        vals = [v for v in vals if v is not None]
        return float(np.mean(vals)) if vals else float('nan')

    k_ms = {k: mean_ms([ev_ms(tm[i][0], tm[i][1]) for tm in timers])
            for i, k in enumerate(('stats_gram', 'project', 'reconstruct'))}
    # what lies BETWEEN the three kernels on the compute stream (same events): the host gap of fit() (download, eigen-solve,
    # upload), the join / release of the field exchange in front of reconstruct, and reconstruct -> the next step's Gram pass
    deferred = spr._defers() and not head_sync
    if deferred:
        # the reconstruct kernel of step k sits between the Gram kernel and the projection of step k + 1 (its event pair is the
        # one armed in step k + 1)
        gaps_ms = dict(gram_to_reconstruct=mean_ms([ev_ms(tm[0][1], tm[2][0]) for tm in timers[1:]]),
                       reconstruct_to_project=mean_ms([ev_ms(tm[2][1], tm[1][0]) for tm in timers[1:]]),
                       gram_to_project=mean_ms([ev_ms(tm[0][1], tm[1][0]) for tm in timers]),
                       project_to_next_gram=mean_ms([ev_ms(a[1][1], b[0][0]) for a, b in zip(timers, timers[1:])]))
    else:
        gaps_ms = dict(gram_to_project=mean_ms([ev_ms(tm[0][1], tm[1][0]) for tm in timers]),
                       project_to_reconstruct=mean_ms([ev_ms(tm[1][1], tm[2][0]) for tm in timers]),
                       reconstruct_to_next_gram=mean_ms([ev_ms(a[2][1], b[0][0]) for a, b in zip(timers, timers[1:])]))
    r = spr.r
    # per-launch algorithmic work of each kernel on THIS rank's shard
    alg = {
        'stats_gram': dict(bytes=n_loc * m * B + n_loc * 8, flops=float(n_loc) * m * m),
        'project': dict(bytes=n_loc * m * B + n_loc * r * B, flops=2.0 * n_loc * m * r),
        'reconstruct': dict(bytes=n_loc * r * B + 2 * n_loc * 8, flops=2.0 * n_loc * r),
    }
    def phase_of(k, ms):
        """achieved rates of one kernel, its binding bound (the larger of t_bytes and t_flops at the SPEC peaks)
        and the fraction of that bound's spec peak and of the peak measured on this chip"""
        gbs, tfs = alg[k]['bytes'] / ms / 1e6, alg[k]['flops'] / ms / 1e9
        mfma_bound = alg[k]['flops'] / (MFMA_F64_PEAK_TF * 1e12) > alg[k]['bytes'] / (HBM_PEAK_GBS * 1e9)
        ach, spec, meas = (tfs, MFMA_F64_PEAK_TF, MFMA_F64_MEASURED_TF) if mfma_bound else (gbs, HBM_PEAK_GBS, HBM_MEASURED_GBS)
        return dict(ms=round(ms, 4), GBs=round(gbs, 1), TFLOPs=round(tfs, 3), bound='mfma' if mfma_bound else 'hbm',
                    frac_spec=round(ach / spec, 4), frac_measured_peak=round(ach / meas, 4),
                    frac_hbm_spec=round(gbs / HBM_PEAK_GBS, 4), frac_mfma_spec=round(tfs / MFMA_F64_PEAK_TF, 4))

    phases = {k: phase_of(k, k_ms[k]) for k in k_ms}
    phases['peaks'] = dict(hbm_spec_GBs=HBM_PEAK_GBS, hbm_measured_GBs=HBM_MEASURED_GBS, mfma_f64_spec_TFLOPs=MFMA_F64_PEAK_TF,
                           mfma_f64_measured_TFLOPs=MFMA_F64_MEASURED_TF)

    def roofline_of(kms):
        dom = max(kms, key=kms.get)
        t_bytes = alg[dom]['bytes'] / (HBM_PEAK_GBS * 1e9)
        t_flops = alg[dom]['flops'] / (MFMA_F64_PEAK_TF * 1e12)
        if t_flops > t_bytes:
            rf = dict(kernel=dom, bound='mfma', achieved=round(alg[dom]['flops'] / kms[dom] / 1e9, 3),
                      peak=MFMA_F64_PEAK_TF, unit='TFLOP/s')
        else:
            rf = dict(kernel=dom, bound='hbm', achieved=round(alg[dom]['bytes'] / kms[dom] / 1e6, 1),
                      peak=HBM_PEAK_GBS, unit='GB/s')
        rf['frac'] = round(rf['achieved'] / rf['peak'], 4)
        rf['ms'] = round(kms[dom], 4)
        return rf

    roof = roofline_of(k_ms)
    dom = roof['kernel']
    roof['traffic'] = None                           # HBM bytes/launch from separate rocprofv3 --pmc passes
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            pm = json.load(f).get(args.workload if not share else f'{args.workload}/{share}')
        ent = (pm or {}).get(dom) if pm and 'kernel' not in pm else (pm if dom == 'stats_gram' else None)
        if ent and world == 1:
            roof['traffic'] = ent['hbm_bytes']
            roof['algorithmic_bytes'] = int(alg[dom]['bytes'])
            roof['traffic_source'] = ('profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE, '
                                      f"separate passes (round {ent.get('round', '?')})")
    except (OSError, ValueError):
        pass
    # ONE line per rank that says why a run is slow (every rank logs its own; rank 0 also carries all of them in the JSON
    # line and names the slowest): its own step time (before it waited for the others), the three kernels, the host
    # eigen-solve and the collectives of the headline loop
    nan = float('nan')
    mine = dict(step_ms=1e3 * dt_own / args.steps, gram_ms=k_ms['stats_gram'], project_ms=k_ms['project'],
                reconstruct_ms=k_ms['reconstruct'], eig_ms=float(getattr(spr, 'eig_ms_', nan)),
                allreduce_ms=comm_main.get('allreduce') if comm_main.get('allreduce') is not None else nan,
                gather_ms=(comm_sync or comm_main).get('gather') if (comm_sync or comm_main).get('gather') is not None else nan,
                gather_exposed_ms=comm_main.get('gather_exposed') if comm_main.get('gather_exposed') is not None else nan)
    log(f'[rank {rank}] ' + ' '.join(f'{k_}={v:.3f}' for k_, v in mine.items()))
    per_rank = None
    slowest = None
    if world > 1:
        keys = list(mine)
        kt = torch.tensor([mine[k_] for k_ in keys], dtype=torch.float64, device=eng.device)
        allk = [torch.empty_like(kt) for _ in range(world)]
        dist.all_gather(allk, kt)
        per_rank = []
        for q, t in enumerate(allk):
            v = dict(zip(keys, t.tolist()))
            rf = roofline_of(dict(stats_gram=v['gram_ms'], project=v['project_ms'], reconstruct=v['reconstruct_ms']))
            rf['rank'] = q
            rf['kernels_ms'] = dict(stats_gram=round(v['gram_ms'], 4), project=round(v['project_ms'], 4),
                                    reconstruct=round(v['reconstruct_ms'], 4))
            rf['timeline_ms'] = {k_: (None if x != x else round(x, 4)) for k_, x in v.items()}
            per_rank.append(rf)
        q_slow = int(np.argmax([e['timeline_ms']['step_ms'] for e in per_rank]))
        slowest = dict(rank=q_slow, **per_rank[q_slow]['timeline_ms'])
        if rank == 0:
            log(f'[rank 0] slowest rank: {q_slow} ({per_rank[q_slow]["timeline_ms"]})')
    # whole-step algorithmic bytes: (2m + 2r) n B for X and Ur + 8 n (field, f64) + 16 n (row means)
    step_bytes = (2 * m + 2 * r) * float(n_job) * B + 24.0 * n_job
    hbm_frac = step_bytes / (dt / args.steps) / (world * HBM_PEAK_GBS * 1e9)

    def timed3(fn):
        """median wall time of three calls, each bracketed by barrier + device sync"""
        ts = []
        for _ in range(3):
            barrier(); t_a = time.perf_counter()
            fn()
            barrier(); ts.append(time.perf_counter() - t_a)
        return 1e3 * sorted(ts)[1]

    # The rest of the path, in every run ("timed and reported as separate lines"): optimal_placement('qr'),
    # train, predict.  Two untimed calls first: code objects, allocator growth and the interpreter's first full GC pass
    # all land in the first two placements (tools/placement_probe.py: 180 / 105 / 55 / 55 / 55 ms at config 3)
    wd.beat('placement / train / predict')
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
    gap_min = float(spr.pivot_gap_.min())
    f32_basis = f32
    path = dict(optimal_placement_ms=round(t_place, 3), train_ms=round(t_train, 3), predict_ms=round(t_pred, 3),
                pivot_sweeps=int(spr.pivot_sweeps_), pivot_pool_sweeps=int(spr.pivot_pool_sweeps_),
                min_pivot_gap=gap_min, placement_from_row_norms=bool(spr.placement_from_norms_),
                placement_GBs_of_basis=round(n_loc * r * B / t_place / 1e6, 1),
                sensors_crc32=int(__import__('zlib').crc32(np.ascontiguousarray(spr.sensors_, dtype=np.int64).tobytes())),
                timing='median of 3 calls after 2 warm-up calls, barrier + device sync on both sides')
    if f32_basis:
        # f32-stored basis (config 5): the sensors are those of the STORED basis; every pick led its runner-up by at least
        # min_pivot_gap (relative), to be read against the rounding of an f32 entry, 2^-24 = 6e-8.  Equality with the
        # sensors of the f64 basis (the reference's) is CHECKED on the CPU-sized sample (parity.sensors_equal) and at full
        # size INFERRED from this margin -- a property of the data's spectrum, not of the code.
        path['min_pivot_gap_over_f32_rounding'] = round(gap_min / 2.0 ** -24, 1)
        path['sensors_vs_f64_basis'] = 'inferred from min_pivot_gap at full size; checked exactly on the parity sample'

    extra = {}
    wd.beat('extra')
    if args.extra:
        wd.stop()
        extra = dict(optimal_placement_ms=path['optimal_placement_ms'], train_ms=path['train_ms'],
                     predict_ms=path['predict_ms'], min_pivot_gap=gap_min, pivot_sweeps=path['pivot_sweeps'],
                     timing='median of 3 calls after 2 warm-up calls')
        extra['placement_from_row_norms'] = bool(spr.placement_from_norms_)
        # A/B of ROM.placement_norms (the squared row norms fit() leaves for the placement): what the projection pays for
        # writing them, what the placement saves by not reading the basis for them.  Default "auto" first, then off.
        def fit_only():
            spr.fit(select_modes='number', n_modes=s)
        t_fit = timed3(fit_only)
        spr.placement_norms = False
        fit_only()
        t_fit_off = timed3(fit_only)
        spr.optimal_placement()
        t_place_off = timed3(spr.optimal_placement)
        assert not spr.placement_from_norms_ and np.array_equal(piv_first, spr.sensors_), 'placement differs without norms'
        extra.update(fit_ms=round(t_fit, 3), fit_without_row_norms_ms=round(t_fit_off, 3),
                     optimal_placement_without_row_norms_ms=round(t_place_off, 3),
                     pivot_sweeps_without_row_norms=int(spr.pivot_sweeps_))
        del spr.placement_norms                                # back to the class default
        fit_only()
        # The reference's output contract: reconstruct() returns a HOST (n, n_p) ndarray (:371-375).  One vector and eight;
        # the field leaves in row chunks whose D2H copies overlap the next chunk's kernel (engine.reconstruct_to_host)
        a_host = spr.Ar[:1].copy()
        A8 = np.vstack([a_host * (1.0 + 0.01 * j) for j in range(8)])
        x1 = spr.reconstruct(a_host)
        assert x1.shape == (n_glob if not share else n_loc, 1)
        t_h1 = timed3(lambda: spr.reconstruct(a_host))
        del x1
        x8 = spr.reconstruct(A8)
        nb1 = float(x8.shape[0]) * 8
        del x8                                         # (held across the timed calls, it and the next result exceed the page-locked
        t_h8 = timed3(lambda: spr.reconstruct(A8))     #  budget of 8 GiB at config 3 and the timed calls fall to pageable copies: 17 GB/s)
        extra.update(reconstruct_to_host_ms=round(t_h1, 3), reconstruct_to_host_GBs=round(nb1 / t_h1 / 1e6, 1),
                     reconstruct_to_host_8_ms=round(t_h8, 3), reconstruct_to_host_8_GBs=round(8 * nb1 / t_h8 / 1e6, 1),
                     reconstruct_device_only_ms=round(k_ms['reconstruct'], 3))
        # ... and its input contract: X handed over as a host ndarray (:74); rate of the upload on a 2 GB sample
        hs = np.ones((min(n_loc, 4_000_000), 64))
        eng.to_device(hs[:1024])
        barrier(); t_a = time.perf_counter()
        hd = eng.to_device(hs)
        barrier(); t_up = time.perf_counter() - t_a
        extra.update(upload_GBs=round(hs.nbytes / t_up / 1e9, 1), upload_sample_GB=round(hs.nbytes / 1e9, 2))
        del hd, hs

    cpu = None
    parity = None
    wd.beat('cpu baseline / parity leg')
    if rank == 0 and world == 1 and not share and not args.no_cpu:
        from oracle import spr_oracle as orc
        # The sample is sized by a TIME budget, not a fixed cell count: the oracle is timed on a small probe
        # (which also warms the BLAS pool up), its rate -- fit + reconstruct are O(n) at fixed m, r -- gives the number of cells
        # that fill SPR_BENCH_CPU_BUDGET_S seconds (default 30), capped by the workload and by half of the host's free RAM
        # (the oracle holds X, X0, U and the field: about 6 copies of the sample)
        budget_s = float(os.environ.get('SPR_BENCH_CPU_BUDGET_S', '30'))
        probe_cells = min(n_points, max(2000, wl['cpu_cells'] // 5))

        def sample_of(cells):
            ix = torch.cat([torch.arange(f * n_points, f * n_points + cells, device=eng.device) for f in range(F)])
            return eng.to_host(Xd[ix])              # first `cells` cells of every feature: a valid (cells*F) x m problem

        cc = probe_cells
        probe_note = ''
        if probe_cells < n_points:
            Xp = sample_of(probe_cells)
            wd.stop()
            t_p = time.perf_counter()
            orc.fit_reconstruct_timed(Xp.astype(np.float64) if f32 else Xp, F, s)
            t_p = time.perf_counter() - t_p
            cc = int(probe_cells * budget_s / max(t_p, 1e-3))
            try:
                import psutil
                cc = min(cc, int(0.5 * psutil.virtual_memory().available / (6.0 * F * m * 8)))
            except ImportError:
                pass
            cc = max(probe_cells, min(n_points, cc // 1000 * 1000))
            probe_note = f'; sample sized for a {budget_s:.0f} s budget from a {probe_cells}-cell probe ({t_p:.2f} s)'
            del Xp
        Xs = sample_of(cc)
        wd.stop()                                   # the oracle runs for 10-30 s of host time; nothing below can hang on a peer
        t1 = time.perf_counter()
        # an f32-stored sample goes to the oracle widened: the reference itself would run in f32 on it
        Xo = Xs.astype(np.float64) if f32 else Xs
        xr_cpu, st_cpu = orc.fit_reconstruct_timed(Xo, F, s)
        t_cpu = time.perf_counter() - t1
        blas, cores, note = 'unknown', 1, ''
        try:
            import threadpoolctl
            info = [p for p in threadpoolctl.threadpool_info() if p.get('user_api') == 'blas']
            # the oracle's time is LAPACK/BLAS time: `cores` = the threads that BLAS pool really ran with
            cores = max([p.get('num_threads', 1) for p in info] or [1])
            blas = '; '.join(sorted({f"{p.get('internal_api')} {p.get('version')} ({p.get('threading_layer', p.get('user_api'))}, "
                                      f"{p.get('num_threads')} threads)" for p in info})) or blas
        except Exception:
            pass
        phys = _physical_cores()
        if phys and cores < phys:
            note = (f'; the BLAS build caps its pool at {cores} threads: {phys - cores} of the {phys} physical cores idle')
        extrap = '' if cc == n_points else (f'; rate measured on this sample -- for the full {n_points}-cell workload it is an '
                                            'EXTRAPOLATION, linear in n (fit and reconstruct are O(n) at fixed m, r)')
        cpu = dict(value=round(Xs.nbytes / t_cpu / 1e9, 4), unit='GB/s', cores=int(cores), kind='port',
                   cpu_model=_cpu_model(), host_cpus=os.cpu_count(), physical_cores=phys, blas=blas,
                   sample=f'{cc} cells x {F} features x {m} snapshots ({Xs.nbytes / 1e6:.0f} MB), s={s}: '
                          f'oracle fit+reconstruct {t_cpu:.2f} s (same generator, first {cc} cells per feature)'
                          + extrap + probe_note + note)
        # parity on the same sample: GPU path vs oracle.  The oracle ran in f64 on the stored values widened, which is
        # what the reference computes for a float32 X as well (X_cnt / X_scl are float64: X0, U float64, :106-107, :169)
        Xs_in = DeviceMatrix(eng.to_device(Xs, dtype=torch.float32), basis='f32') if f32 else Xs
        sp2 = SPR(Xs_in, F, None, engine=eng)
        sp2.fit(select_modes='number', n_modes=s)
        sg = np.sign(np.sum(sp2.Ur.astype(np.float64) * st_cpu['Ur'], axis=0))   # same coefficients need the same signs
        xr_gpu = sp2.reconstruct(st_cpu['Ar'][0] * sg)
        sp2.optimal_placement()
        # sensors: against the REFERENCE's choice -- dgeqp3's pivots of the oracle's f64 basis -- also when the device
        # basis is stored in f32 (config 5); min_pivot_gap says how far every pick was from a tie (f32 rounding: 6e-8)
        piv_cpu, _ = orc.qr_pivots(st_cpu['Ur'])
        parity = dict(sensors_equal=bool(np.array_equal(sp2.sensors_, piv_cpu)),
                      sensors_vs='dgeqp3 pivots of the oracle f64 basis (the reference\'s sensors)',
                      basis_dtype=str(sp2.Ur.dtype), min_pivot_gap=float(sp2.pivot_gap_.min()),
                      field_rel_fro=float(np.linalg.norm(xr_gpu - xr_cpu) / np.linalg.norm(xr_cpu)),
                      sigma_rel=float(np.max(np.abs(sp2.Sigma_r - st_cpu['Sigma_r']) / st_cpu['Sigma_r'])),
                      sigma1_over_sigmas=float(sp2.Sigma_r[0] / sp2.Sigma_r[-1]))

    if rank == 0:
        store = 'f32 storage / f64 arithmetic' if f32 else 'f64'
        if share:
            what = (f"{args.workload}: rank {args.share_rank}'s block of a {share}-rank run ({n_loc} rows of {n_glob}), "
                    f'alone on one GPU with its collectives in a 1-rank RCCL group -- NOT an N-GPU number')
        elif plan['scaling'] == 'strong':
            what = (f"{args.workload}: {n_points} cells total x {F} features x {m} snapshots, {s} modes/sensors, {store}, "
                    f'row-sharded over {world} GPU(s) ({n_loc} rows each)')
        else:
            what = (f"{args.workload}: {wl['cells']} cells/GPU x {F} features x {m} snapshots, {s} modes/sensors, {store}, "
                    f'rows sharded over {world} GPU(s)')
        out = {
            'metric': 'SPR fit+reconstruct throughput (GB/s snapshot matrix)', 'value': round(value, 2),
            'unit': 'GB/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': plan['scaling'],
            'vs_baseline': None, 'dtype': 'f64' if not f32 else 'f64 arithmetic on f32-stored data', 'data': 'synthetic',
            'config': {'workload': what, 'rows_per_gpu': n_loc, 'rows_total': n_job,
                       'snapshot_GB_per_gpu': round(n_loc * m * B / 1e9, 3), 'storage': 'f32' if f32 else 'f64',
                       **({'placement_norms': args.placement_norms} if args.placement_norms != 'auto' else {})},
            'hbm_roofline_frac_step': round(hbm_frac, 4),
            # the same K steps with the field all-gather joined inside every step (N > 1; at N = 1 there is no gather)
            'ms_per_step_sync_gather': (round(1e3 * dt_sync / args.steps, 4) if dt_sync is not None
                                        else round(ms_per_step, 4)),
            'headline_loop': ('sync gather' if (args.sync_gather or not dist_on) else
                              f'field exchange ({head_path}) left in flight under the next Gram pass'),
            'comm': comm,
            **({'refine': dict(gram_refine_passes=int(spr.gram_refine_passes_), refine_ms=round(float(spr.refine_ms_), 3),
                               refine_over_plain_gram=round(float(spr.refine_ms_) / k_ms['stats_gram'], 3),
                               refine_device_ms=round(spr.refine_profile_['device_ms'], 3),
                               refine_host_ms=round(spr.refine_profile_['host_ms'], 3),
                               refine_host_pieces={k: round(v, 3) for k, v in spr.refine_profile_.items() if k.endswith('_ms') and k not in ('device_ms', 'host_ms')},
                               sigma1_over_sigma_s=float(spr.Sigma_r[0] / spr.Sigma_r[-1]),
                               note='fit() above sigma_1/sigma_r = 1e4: one more pass over X per refinement pass -- 2 n m^2 '
                                    'flops of projection onto all m first-stage vectors + n m^2 of Gram, 3x the plain Gram '
                                    'pass by flop count; refine_ms is host wall time of the whole refinement, inside ms_per_step')}
               if getattr(spr, 'gram_refine_passes_', 0) else {}),
            'placement_ms': path['optimal_placement_ms'], 'train_ms': path['train_ms'], 'predict_ms': path['predict_ms'],
            'pivot_sweeps': path['pivot_sweeps'], 'min_pivot_gap': path['min_pivot_gap'], 'path': path,
            **({'p2p_loopback': f'{args.p2p_loopback} imaginary peers inside this GPU: the pushes, counters and waits of a '
                                f'{args.p2p_loopback + 1}-rank p2p field exchange with both ends in this HBM -- NOT an N-GPU number'}
               if args.p2p_loopback else {}),
            **({'rehearsal': f'{backend} backend, all ranks on one GPU: exercises the code path, measures nothing'}
               if (backend != 'nccl' or os.environ.get('SPR_BENCH_ONE_GPU') == '1') else {}),
            'roofline': roof, 'cpu_baseline': cpu, 'phases': phases, 'parity': parity,
        }
        if f32:
            # BASELINE.md 4 counts the field write as n B bytes (B = 4 for f32 storage); the product writes the field in float64
            # like the reference (Ur @ Ar.T and unscale_data are float64 for any dtype of X): 8 n here, 4 n more than the model
            out['byte_model_note'] = ('hbm_roofline_frac_step counts the reconstructed field as float64 (8 bytes per row, what the product '
                                      'writes and the reference returns); BASELINE.md section 4 models it with B = 4 for this storage: '
                                      f'{4.0 * n_job / 1e9:.2f} GB of {step_bytes / 1e9:.1f} GB per step more than that model')
        out['rank_timeline_ms'] = {k_: (None if v != v else round(v, 4)) for k_, v in mine.items()}
        out['gaps_ms'] = {k_: (None if (v is None or v != v) else round(v, 4)) for k_, v in gaps_ms.items()}
        out['reconstruct_launch'] = ("deferred (ROM.defer_reconstruct, the library's default for reconstruct(to_host=False, wait=False)): "
                                     "step k's reconstruct kernel is enqueued in the host gap of step k + 1's fit(); the last one is "
                                     "flushed inside the timed region" if deferred else "behind the step's own projection")
        if dt_pipe is not None:
            # N = 1: the same K steps with reconstruct(to_host=False, wait=False) -- the asynchronous form, deferred into the next
            # fit()'s host gap unless --no-defer-reconstruct -- every step inside the timed region, the last one flushed in it
            out['ms_per_step_pipelined'] = round(1e3 * dt_pipe / args.steps, 4)
            out['value_pipelined'] = round(x_bytes / (dt_pipe / args.steps) / 1e9, 2)
            out['pipelined_loop'] = ('reconstruct(to_host=False, wait=False): ' + ('launch deferred into the next fit()\'s host gap '
                                     '(ROM.defer_reconstruct, default)' if spr._defers() else 'launched behind its own projection'))
        if ballast is not None:
            out['ballast_GB'] = round(ballast.numel() / 1e9, 2)
        free_b, total_b = torch.cuda.mem_get_info(eng.device)
        px_ = spr.__dict__.get('_p2p')
        out['hbm'] = dict(total_GB=round(total_b / 1e9, 2), free_at_end_GB=round(free_b / 1e9, 2),
                          torch_peak_allocated_GB=round(torch.cuda.max_memory_allocated() / 1e9, 2),
                          torch_peak_reserved_GB=round(torch.cuda.max_memory_reserved() / 1e9, 2),
                          p2p_buffers_GB=round(getattr(px_, 'total_bytes', 0) / 1e9, 2) if px_ is not None and px_.base else 0.0,
                          note='free/total: hipMemGetInfo on this rank at the end of the run (everything of the process and of RCCL); the p2p '
                               'buffers come from hipMalloc, outside torch\'s allocator')
        out['compute_stream'] = 'own (non-default) stream' if own_stream else 'default stream'
        out['hw_queues'] = os.environ.get('GPU_MAX_HW_QUEUES', 'runtime default (4)')
        out['gap_filler'] = dict(on=bool(spr.gap_filler), rows_per_fit=(int(np.mean(fill_rows)) if fill_rows else 0),
                                 note='opt-in (--gap-filler): fit() re-queues its Gram kernel over this many rows of X into the '
                                      'host gap and discards the result -- work inside the timed step that the GB/s model does '
                                      'not count')
        if per_rank:
            out['roofline_per_rank'] = per_rank
            out['slowest_rank'] = slowest
        if extra:
            out['extra'] = extra
        out['peak_hbm_GB'] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if world > 1 or force_dist:
        dist.destroy_process_group()


def _physical_cores():
    try:
        seen = set()
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        return len(seen) or None
    except OSError:
        return None


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


if __name__ == '__main__':
    main()
