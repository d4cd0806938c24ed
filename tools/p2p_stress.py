#!/usr/bin/env python3
"""GPU box (round 5): the step loop of bench.py -- fit, join of the previous field, reconstruct left in flight -- repeated many hundred
times over the p2p exchange in loopback (imaginary peers inside this GPU: the real pushes, counters and waits), with and without
ROM.defer_reconstruct; every joined field and every imaginary peer's copy of it is compared bit for bit with the first round's.
usage: python tools/p2p_stress.py [rounds] [peers]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
peers = int(sys.argv[2]) if len(sys.argv) > 2 else 7
os.environ['SPR_P2P_LOOPBACK'] = str(peers)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29611')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from openmeasure_amd.engine import HipEngine  # noqa: E402
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix, RowShard  # noqa: E402
from openmeasure_amd.synth import make_R  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group('gloo')
eng = HipEngine('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(eng.device))
wl = bench.WORKLOADS['c4s']
plan = bench.shard_plan(wl, 8, 3)
R = eng.to_device(make_R(wl['m'], wl['s'], seed=1234))
Xd = eng.synth(plan['n_loc'], wl['m'], plan['row0'], plan['n_points'], R, 1e-3, 1234)
bad = 0
for defer in (False, True):
    spr = SPR(DeviceMatrix(Xd), wl['features'], None, engine=eng,
              shard=RowShard(plan['row0'], plan['n_glob'], force_collectives=True, partial=True, gather='p2p'))
    spr.defer_reconstruct = defer
    spr.fit(select_modes='number', n_modes=wl['s'])
    a = [eng.to_device(spr.Ar[:1] * (1.0 + 0.5 * j)) for j in range(3)]
    refs = [spr.reconstruct(aj, to_host=False, wait=True).clone() for aj in a]
    px = spr._p2p
    prev, prev_j = None, None
    for it in range(rounds):
        spr.fit(select_modes='number', n_modes=wl['s'])
        if prev is not None:
            f = prev.wait()
            if not torch.equal(f, refs[prev_j]):
                bad += 1
            if it % 50 == 1:                                   # the imaginary peers' copies hold the same block
                torch.cuda.synchronize()
                n_p, n_tot = px.shape
                for q in px.peers:
                    off = q * px.n_buf * px.field_bytes
                    peer = px._mem[off:off + n_p * n_tot * 8].view(torch.float64).view(n_p, n_tot)
                    if not torch.equal(peer, refs[prev_j]):
                        bad += 1
        prev_j = it % 3
        prev = spr.reconstruct(a[prev_j], to_host=False, wait=False)
    if not torch.equal(prev.wait(), refs[prev_j]):
        bad += 1
    px.check()
    print(f'defer_reconstruct={defer}: {rounds} rounds, {peers} imaginary peers, gather path {spr.gather_path_[:3]}, differing fields so far: {bad}')
    spr.close()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
