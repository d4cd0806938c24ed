"""bench.py's N-rank launcher and shard arithmetic, driven on the CPU (BASELINE configs 4/5).

`python bench.py --gpus N` without a torchrun environment starts the N rank processes itself; here the same
launcher starts N gloo ranks that run the product's sharded host logic with the NumPy test double, so the process
spawning, the rank environment, the relay of rank 0's JSON line, failure propagation and the strong-scaling shard
arithmetic (features straddling the shards) are covered without a GPU."""
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CHILD = os.path.join(ROOT, 'tests', '_bench_rank_double.py')


def test_default_workload_and_scaling():
    a1 = bench.parse_args([])
    assert (a1.gpus, a1.workload) == (1, 'c3')
    a8 = bench.parse_args(['--gpus', '8'])
    assert a8.workload == 'c4'
    # round 6: the headline of an N > 1 line is the exchange the LIBRARY chose; deferred launches are the library's default (the old
    # flag is still accepted, --no-defer-reconstruct switches them off); --ballast-gb is a developer aid and off
    assert a8.headline_gather == 'library' and a8.gather == 'auto' and not a8.no_defer_reconstruct and a8.ballast_gb == 0.0
    assert bench.parse_args(['--defer-reconstruct']).defer_reconstruct and bench.parse_args(['--no-defer-reconstruct']).no_defer_reconstruct
    assert bench.parse_args(['--workload', 'c5', '--share-of', '8', '--p2p-loopback', '7', '--ballast-gb', '10']).ballast_gb == 10.0
    assert bench.parse_args(['--gpus', '8', '--workload', 'c5']).workload == 'c5'
    wl = bench.WORKLOADS['c4']
    for world in (1, 2, 4, 8):
        plans = [bench.shard_plan(wl, world, r) for r in range(world)]
        assert all(p['scaling'] == 'strong' and p['n_points'] == 10_000_000 and p['n_glob'] == 90_000_000 for p in plans)
        assert [p['row0'] for p in plans] == [r * (90_000_000 // world) for r in range(world)]
        assert sum(p['n_loc'] for p in plans) == 90_000_000
    # config 4 at N = 1 is config 3
    assert bench.shard_plan(wl, 1, 0) == dict(bench.shard_plan(bench.WORKLOADS['c3'], 1, 0), scaling='strong')
    # 8 ranks: 11.25M rows each, so rank 0 holds all of feature 0 and the first 1.25M rows of feature 1
    p = bench.shard_plan(wl, 8, 0)
    assert p['n_loc'] == 11_250_000 and p['n_loc'] > p['n_points']
    # weak scaling: config 5 is 6.25M cells per GPU, 50M cells on 8
    p5 = bench.shard_plan(bench.WORKLOADS['c5'], 8, 7)
    assert p5['scaling'] == 'weak' and p5['n_points'] == 50_000_000 and p5['n_loc'] == 100_000_000
    assert p5['row0'] == 700_000_000
    # a world that does not divide the cells: cells rounded down, shards equal
    p3 = bench.shard_plan(dict(cells=100, features=3, scaling='strong'), 7, 6)
    assert p3['n_points'] == 98 and p3['n_loc'] * 7 == p3['n_glob'] == 294


@pytest.mark.parametrize('world', [2, 3])
def test_launcher_runs_sharded_job_over_gloo(tmp_path, world):
    out = io.StringIO()
    field_path = str(tmp_path / 'field.npy')
    rc = bench.launch_ranks(world, [sys.executable, CHILD, '--gpus', str(world), '--cells', '120', '--out', field_path],
                            timeout=240, relay=out)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # exactly one JSON line, rank 0's
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == world and rec['scaling'] == 'strong'
    assert rec['rows_total'] == 360 and rec['rows_per_gpu'] * world == 360
    # collective timing (bench.py's `comm` object): one all-reduce bracket per fit in both loop forms; the gather is timed
    # issue -> join where it is joined inside the step, and its exposed wait where it is left in flight
    assert rec['brackets_pipelined'] == {'allreduce': 3, 'gather_exposed': 3}
    assert rec['brackets_sync'] == {'allreduce': 3, 'gather': 3}
    comm = rec['comm']
    assert comm['allreduce_ms'] > 0 and comm['gather_ms'] > 0 and comm['gather_exposed_ms'] >= 0
    assert comm['allreduce_bytes'] == (3 * 10 * 10 + world * 3 * 3) * 8 and comm['gather_bytes_per_rank'] == 8 * 360 // world
    # same job in one process with the same engine: the gathered field agrees
    from openmeasure_amd.sparse_sensing import SPR
    from tests._bench_rank_double import make_X
    from tests.numpy_engine import NumpyEngine
    X = make_X(360, 10, 3)
    spr = SPR(X, 3, None, engine=NumpyEngine())
    spr.fit(select_modes='number', n_modes=4)
    ref = spr.reconstruct(spr.Ar[0])
    got = np.load(field_path)
    assert got.shape == ref.shape == (360, 1)
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)


def test_launcher_propagates_a_failed_rank():
    out = io.StringIO()
    rc = bench.launch_ranks(2, [sys.executable, CHILD, '--gpus', '2', '--fail-rank', '1', '--hang-rank', '0'],
                            timeout=240, relay=out)
    assert rc == 3                                           # rank 1's exit code; rank 0 (hanging) was terminated
    assert not [ln for ln in out.getvalue().splitlines() if ln.startswith('{')]


def test_launcher_kills_a_rank_that_ignores_sigterm():
    """A rank stuck in a collective (here: one that ignores SIGTERM) must not hang the launcher: SIGKILL after the grace
    period, non-zero exit code."""
    import time
    out = io.StringIO()
    t0 = time.time()
    rc = bench.launch_ranks(2, [sys.executable, CHILD, '--gpus', '2', '--fail-rank', '1', '--hang-rank', '0', '--deaf'],
                            timeout=240, relay=out, grace=2.0)
    assert rc == 3 and time.time() - t0 < 120
    rc = bench.launch_ranks(2, [sys.executable, CHILD, '--gpus', '2', '--hang-rank', '0', '--deaf'], timeout=20, relay=out,
                            grace=2.0)
    assert rc == 124                                          # timeout: terminate, then kill


def test_launcher_parent_counts_gpus_without_hip():
    """bench.py --gpus N as the launcher never imports torch or touches the HIP runtime: the device count is read from
    sysfs (None on a machine without the driver)."""
    import subprocess
    code = ("import sys, bench; n = bench.count_gpus(); "
            "assert 'torch' not in sys.modules, 'launcher parent imported torch'; print(n)")
    p = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip() == 'None' or int(p.stdout) >= 0


def test_world_size_mismatch_is_an_error():
    import subprocess
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--no-cpu'], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode != 0 and 'WORLD_SIZE=2' in p.stderr


def test_watchdog_ends_a_rank_that_makes_no_progress():
    """bench.Watchdog (round 5): a rank stuck in a collective says where it stands -- phase label, the collective brackets
    recorded so far, every thread's stack -- and ends its process with a non-zero code; a rank that keeps beating is left
    alone."""
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "wd = bench.Watchdog(0.6, rank=5, describe=lambda: dict(collective_brackets={'allreduce': 7}))\n"
            "for _ in range(8):\n    time.sleep(0.2); wd.beat('warm-up')\n"
            "wd.beat('join of the previous field')\ntime.sleep(30)\n" % ROOT)
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3, (p.returncode, p.stderr[-400:])
    assert "WATCHDOG: no progress" in p.stderr and "'join of the previous field'" in p.stderr
    assert "'allreduce': 7" in p.stderr and 'time.sleep(30)' not in p.stdout and 'File "<string>"' in p.stderr
    # stop(): the host-only legs of the run (CPU oracle) are not watched
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "wd = bench.Watchdog(0.3); wd.stop(); time.sleep(1.2); print('alive')\n" % ROOT)
    p2 = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True, timeout=60)
    assert p2.returncode == 0 and 'alive' in p2.stdout
