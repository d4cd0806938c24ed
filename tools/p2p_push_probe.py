#!/usr/bin/env python3
"""GPU box: where the host time of one P2PFieldGather.push() goes (loopback exchange, L imaginary peers, one 90 MB block):
the event record + the copy streams' waits, the library call, and -- inside it -- the launches and copies one by one
(SPR_P2P_PROBE=1 makes spr_field_gather_p2p print its own per-call clock).  python3 tools/p2p_push_probe.py [L] [n_streams]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
BUSY = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(sys.argv) > 2:
    os.environ['SPR_P2P_STREAMS'] = sys.argv[2]
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np
import torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.p2p import P2PFieldGather

eng = HipEngine('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(eng.device))
n_loc = 11_250_000
px = P2PFieldGather(eng, 1, 0, lambda t: t[None], loopback=L)
px.ensure(1, n_loc)
work = torch.empty(64 << 20, device=eng.device)
pieces = dict(begin=[], record=[], waits=[], call=[], join=[])
real = px.lib.spr_field_gather_p2p


def timed_call(*a):
    t0 = time.perf_counter()
    rc = real(*a)
    pieces['call'].append(time.perf_counter() - t0)
    return rc


px.lib.spr_field_gather_p2p = timed_call
for it in range(60):
    t0 = time.perf_counter()
    out = px.begin()
    pieces['begin'].append(time.perf_counter() - t0)
    if BUSY:
        for _ in range(BUSY):                           # a compute stream that is still busy when the push is issued (the bench's case)
            work.mul_(1.0001)
    out[:, :n_loc].fill_(float(it))
    t0 = time.perf_counter()
    k = px.push(0, n_loc)
    t_push = time.perf_counter() - t0
    work.mul_(1.0001)                                   # something for the compute stream while the copies run
    t0 = time.perf_counter()
    px.join(k)
    pieces['join'].append(time.perf_counter() - t0)
    pieces['record'].append(t_push)
    torch.cuda.synchronize()
px.check()
for k_, v in pieces.items():
    if v:
        v = np.array(v[10:]) * 1e3
        print(f'{k_:8s} mean {v.mean():.4f} ms  median {np.median(v):.4f}  min {v.min():.4f}  max {v.max():.4f}')
print('busy launches in front:', BUSY)
print('(record = the whole push(); call = spr_field_gather_p2p alone; peers', L, 'streams', len(px._pool), ')')
