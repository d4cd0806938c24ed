"""Host plumbing of the CU-free field gather (include/spr_hip.h: spr_p2p_*, spr_field_gather_p2p*; csrc/p2p.hip).

``reconstruct()`` of a row-sharded SPR ends with every rank's block of the field on every rank (reference: the (n, n_p)
array of sparse_sensing.py:371-375 is whole on every caller).  Over RCCL that is an all-gather whose device kernel cannot
share a compute unit with the Gram / projection workgroups; here every rank keeps ONE persistent copy of the field in a
buffer the other ranks of the node have mapped (interprocess handles, exchanged once through torch.distributed), the
reconstruct kernel writes the rank's own block straight into it, and the block is pushed into the same place of every peer's
copy by the SDMA engines -- a few copy streams, a 64-bit arrival counter raised behind the copies, awaited by ONE single-wave
kernel on the consumer's stream.  Nothing occupies a compute unit while the bytes move, so the exchange really runs under
the next fit()'s MFMA-bound Gram pass.

PyTorch is plumbing: streams, the device context, and ``torch.distributed`` as the channel for the handle bytes.  Buffers
come from the library (an interprocess handle needs the base pointer of an allocation): one for the field copies (+ a
scratch page the self-test uses), one fine-grained 4 KB page for the counters.

Protocol (counters only ever grow; k = number of gathers issued so far on this object, identical on all ranks because a
gather is a collective call; b = k % n_buf, n_buf = 1 unless asked otherwise):
  release   on entering gather k a rank raises release[me] = k in every peer's counter page (one kernel on its COMPUTE stream --
            behind every kernel that read an earlier field): "the fields I was handed before gather k are dead";
  push      per peer p, on one of the copy streams: wait until release[p] >= k - n_buf + 1 (MY page: p no longer reads what
            its buffer b holds), copy my block into p's buffer b, raise arrive[b][me] = k + 1 in p's page and pushed[p] = k + 1
            in mine;
  join      the consumer's stream waits (one kernel) until arrive[b][p] >= k + 1 and pushed[p] >= k + 1 for every peer p: the
            peers' blocks are here and my own pushes have left (their source is my buffer b).  No event of another hardware
            queue is waited for: behind SDMA copies such waits cost 0.9 ms per step (profiles/r05_p2p_gap_experiments.txt).
The field handed out is a VIEW of buffer b: it stays valid until this rank enters its next gather.  (A second buffer only lets
the peers' COPY streams run a step ahead of a slow rank; no compute stream ever waits for a push, so one buffer is the default.)

Failure (round 6: no silent path to a torn field).  Every wait has a wall-clock exit and leaves (counter, value seen) in
page-locked status words this object reads at every gather (check()).  A join that gives up: RuntimeError naming the counter.  A
push whose wait for the peer's RELEASE gives up (the peer has not entered gather k within RELEASE_TIMEOUT_S: it may still be
reading what its buffer holds) cannot take its copies back -- an SDMA command is unconditional -- so the arrival counters it owes
are raised WITH the poison bit (spr_p2p_poison_bit): the peer's join of that gather and the pusher's own both fail, both ranks
raise, nobody is handed the field.  RELEASE_TIMEOUT_S (default 3 x the join's) is longer than JOIN_TIMEOUT_S on purpose: a peer
that is merely slow makes the pusher's JOIN give up first (the pusher raises; the peer's buffer is never touched while it reads),
and only a peer silent for RELEASE_TIMEOUT_S can be written over -- with the poison behind the copy.  Both are settable:
SPR_P2P_JOIN_TIMEOUT_S (default 600 s, torch.distributed's own default for a collective) and SPR_P2P_RELEASE_TIMEOUT_S.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _lib

_FLAG_BYTES = 4096
_SCRATCH_BYTES = 4096
_REASON_BYTES = 200                  # a rank's verdict and the reason it gives, all-gathered (_agree)
_BUS_ID_BYTES = 24                   # "0000:c1:00.0" and its terminator, with room
# 5 counter arrays of `world` uint64 in the 4 KB counter page (world <= 102) AND a join table of 2 (world - 1) counters in one
# kernel argument of 128 pointers (csrc/p2p.hip kMaxFlags): beyond it the join would be refused mid-gather
_MAX_WORLD = 65                      # (6 counter arrays of 65 words fit the page as well)
_KINDS = {'coarse': 0, 'fine': 1, 'uncached': 2}


class _Raw:
    """A range of device memory as an object torch.as_tensor() can wrap without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = dict(shape=(int(nbytes),), typestr='|u1', data=(int(ptr), False), version=2,
                                             strides=None)


def _ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*[int(p) for p in ptrs])


class P2PUnavailable(RuntimeError):
    """The ranks cannot map each other's buffers (different nodes, no interprocess handles, self-test failed)."""


class P2PFieldGather:
    """One per sharded ROM object; every method that says COLLECTIVE must be called by all ranks."""

    SELFTEST_TIMEOUT_S = 10.0
    JOIN_TIMEOUT_S = 600.0           # the join kernel gives up after this long (status words; see check()); SPR_P2P_JOIN_TIMEOUT_S
    RELEASE_TIMEOUT_S = None         # a push waits this long for the peer's release; None: 3 x the join's; SPR_P2P_RELEASE_TIMEOUT_S
    FIRST_TIMEOUT_S = 20.0           # ... and after this long in the first exchange through new buffers (ROM._p2p_first_exchange)

    def __init__(self, eng, world, rank, all_gather, double_buffer=False, loopback=0):
        # loopback = L (diagnostic, one-rank groups only: bench.py --share-of N --p2p-loopback N-1): L imaginary peers whose
        # copies of the field live behind this rank's own in the same buffer.  The rank then issues exactly the copies,
        # counters and waits of an (L+1)-rank exchange -- L pushes of its block per gather, through the SDMA engines, under
        # whatever runs next -- with both ends in its own HBM.  What one GPU can show of the exchange's cost.
        self.loopback = int(loopback) if int(world) == 1 else 0
        if self.loopback:
            world = 1 + self.loopback
        if world > _MAX_WORLD:
            raise P2PUnavailable(f'{world} ranks exceed the {_MAX_WORLD} one join kernel can wait for')
        for attr, key in (('JOIN_TIMEOUT_S', 'SPR_P2P_JOIN_TIMEOUT_S'), ('RELEASE_TIMEOUT_S', 'SPR_P2P_RELEASE_TIMEOUT_S')):
            env = os.environ.get(key)
            if env:
                v = float(env)
                if not 0.0 < v <= 3600.0:
                    raise ValueError(f'{key}={env}: seconds in (0, 3600]')
                setattr(self, attr, v)
        self.eng, self.world, self.rank = eng, int(world), int(rank)
        self.lib = eng.lib
        self._all_gather = all_gather                          # tensor -> (world, *shape) tensor, COLLECTIVE
        self.n_buf = 2 if double_buffer else 1
        self.k = 0                                             # gathers issued
        self.base = None                                       # my field buffer (device pointer)
        self.fbase = None                                      # my counter page
        self.peer_base, self.peer_fbase = {}, {}               # rank -> mapped pointers
        self.shape = None                                      # (n_p, n_total)
        self.field_bytes = 0
        self.memory = None                                     # 'coarse' | 'uncached': what the field buffers are made of
        torch = eng.torch
        # copy streams: the pushes to different peers are independent and may use different SDMA engines / xGMI links, but a
        # HIP process maps its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and a copy stream that
        # shares a hardware queue with the compute stream puts its SDMA-completion barriers IN FRONT of the next kernel;
        # so few copy streams (SPR_P2P_STREAMS, default 3), the peers dealt round-robin, starting with the next rank
        peers = [q for q in range(self.world) if q != self.rank]
        self.peers = sorted(peers, key=lambda q: (q - self.rank) % self.world)      # rank+1, rank+2, ...: no hot receiver
        n_streams = max(1, min(len(self.peers), int(os.environ.get('SPR_P2P_STREAMS', '3') or 3)))
        self._pool = [torch.cuda.Stream(eng.device) for _ in range(n_streams if self.peers else 0)]
        self.streams = [self._pool[i % n_streams] for i in range(len(self.peers))]
        self.selftest_report = None
        self._ever_allocated = False                           # ensure() has allocated before (identical on all ranks: it is collective)
        # None until the first full-size exchange through these buffers has been compared, block by block, with what the peers
        # say they sent (_shard.py, ROM._p2p_first_exchange); then a short report
        self.verified = None
        self.host_ms = dict(begin=0.0, push=0.0, push_order=0.0, push_call=0.0, join=0.0, calls=0)   # host wall time spent issuing (bench.py reports the means)
        # where a join kernel that gives up leaves (counter index + 1, value seen): page-locked HOST memory the kernel writes
        # directly, so that check() is a plain memory read -- no copy, no synchronisation -- and can run at every gather
        # ... words 0-1: the join; words 2-3: a push whose wait for a peer's release gave up (and the gate of the poison, p2p.hip)
        self._status = torch.zeros(4, dtype=torch.int64).pin_memory()
        self._status_np = self._status.numpy()
        self._poison = int(self.lib.spr_p2p_poison_bit())
        self._tabs = {}                                        # pointer tables of the calls, built once per allocation
        # how the copy streams learn that the block is written: a counter raised on the compute stream (default) or an event of it
        # (SPR_P2P_ORDER=event: A/B)
        self._order_by_counter = os.environ.get('SPR_P2P_ORDER', 'counter') != 'event'

    # ------------------------------------------------------------------ set-up (COLLECTIVE)
    def ensure(self, n_p, n_total):
        """Buffers for an (n_p, n_total) float64 field; (re)allocated and exchanged when the shape grows.  COLLECTIVE when
        it allocates -- every rank sees the same shapes, so every rank takes the same branch.  The field buffers are plain
        device memory when the self-test shows that what a peer writes is seen behind the join (it re-reads lines it has
        read before), uncached device memory otherwise (SPR_P2P_MEMORY=coarse|uncached skips the first / goes straight
        to the second)."""
        n_p, n_total = int(n_p), int(n_total)
        need = -(-n_p * n_total * 8 // 4096) * 4096
        if self.base is not None and need <= self.field_bytes:
            self.shape = (n_p, n_total)
            return
        want = os.environ.get('SPR_P2P_MEMORY')
        if want not in (None, '', 'coarse', 'uncached'):
            raise ValueError("SPR_P2P_MEMORY: 'coarse' or 'uncached'")
        kinds = [want] if want else ['coarse', 'uncached']
        why = ''
        for i, kind in enumerate(kinds):
            # COLLECTIVE teardown of what an earlier allocation / attempt left: the ranks must meet here whether or not THIS rank
            # has anything to free (an allocation that failed on some ranks only -- ADVICE r05 -- would otherwise leave those
            # ranks one collective ahead of the others); every rank takes this branch with the same (i, had buffers before) history
            self.close(collective=True, rendezvous=(i > 0 or self._ever_allocated))
            ok, why = self._allocate(need, kind)
            if ok:
                self.shape = (n_p, n_total)
                self.k = 0
                self.memory = kind
                self.verified = None
                return
        self.close(collective=False)
        raise P2PUnavailable(why)

    def _allocate(self, need, kind):
        """-> (ok on ALL ranks, reason).  COLLECTIVE."""
        torch = self.eng.torch
        self._ever_allocated = True
        self._tabs = {}
        self.field_bytes = need
        copies = self.n_buf * (1 + self.loopback)
        total = copies * need + _SCRATCH_BYTES
        self._scratch = copies * need                          # offset of the scratch page in the field buffer
        hb = int(self.lib.spr_p2p_handle_bytes())
        h_field, h_flags = (C.c_ubyte * hb)(), (C.c_ubyte * hb)()
        ok, why = True, ''
        try:
            base, fbase = C.c_void_p(), C.c_void_p()
            rc = self.lib.spr_p2p_alloc(total, _KINDS[kind], C.byref(base), h_field)
            if rc != 0:
                # the buffers come from hipMalloc, outside PyTorch's allocator, which keeps what its tensors have freed: give
                # that back to the runtime and ask once more (config 5 at N = 8 leaves a few GB next to the shard and the basis)
                torch.cuda.synchronize(self.eng.device)
                torch.cuda.empty_cache()
                rc = self.lib.spr_p2p_alloc(total, _KINDS[kind], C.byref(base), h_field)
            _lib.check(rc, 'spr_p2p_alloc')
            self.base, self.total_bytes = int(base.value), total
            _lib.check(self.lib.spr_p2p_alloc(_FLAG_BYTES, _KINDS['fine'], C.byref(fbase), h_flags), 'spr_p2p_alloc')
            self.fbase = int(fbase.value)
            self._mem = torch.as_tensor(_Raw(self.base, total), device=self.eng.device)          # uint8 view of my field buffer
            self._flags = torch.as_tensor(_Raw(self.fbase, _FLAG_BYTES), device=self.eng.device).view(torch.int64)
            self._mem[self._scratch:].zero_()
            self._flags.zero_()
            torch.cuda.synchronize(self.eng.device)
        except Exception as exc:                               # noqa: BLE001 -- "not available", decided together below
            ok, why = False, f'rank {self.rank}: {exc}'
        if self.loopback:
            if ok:
                for q in self.peers:                           # imaginary peer q: the q-th copy behind mine, my own counters
                    self.peer_base[q] = self.base + q * self.n_buf * need
                    self.peer_fbase[q] = self.fbase
            return ok, why
        # every rank's handle bytes, the GPU it is on and whether it has buffers at all: one exchange, so that all ranks go on
        # or give up together
        mine = np.zeros(2 * hb + 8 + _BUS_ID_BYTES, dtype=np.uint8)
        mine[:hb] = np.frombuffer(bytes(h_field), dtype=np.uint8)
        mine[hb:2 * hb] = np.frombuffer(bytes(h_flags), dtype=np.uint8)
        if ok:
            try:
                bus = C.create_string_buffer(_BUS_ID_BYTES)
                _lib.check(self.lib.spr_p2p_device_id(bus, _BUS_ID_BYTES), 'spr_p2p_device_id')
                mine[2 * hb + 8:] = np.frombuffer(bus.raw, dtype=np.uint8)
            except Exception as exc:                           # noqa: BLE001
                ok, why = False, f'rank {self.rank}: {exc}'
        mine[2 * hb] = 1 if ok else 0
        allh = self.eng.to_host(self._all_gather(torch.tensor(mine, device=self.eng.device)))      # (world, 2 hb + 8 + id)
        if not allh[:, 2 * hb].all():                          # somebody has no buffers: every rank learns why
            return self._agree(ok, why, 'could not allocate exportable buffers')
        try:
            for q in self.peers:
                # BEFORE anything of peer q is mapped: may my GPU write to the GPU q is on?  (A write into mapped memory of a
                # device without peer access -- or one this process cannot see -- is a memory fault, not an error code.)
                bus_q = bytes(allh[q, 2 * hb + 8:].astype(np.uint8)).split(b'\0')[0]
                can = C.c_int32(0)
                _lib.check(self.lib.spr_p2p_peer_access(C.c_char_p(bus_q), C.byref(can)), 'spr_p2p_peer_access')
                if not can.value:
                    raise P2PUnavailable(f"the GPU of rank {q} ({bus_q.decode(errors='replace')}) is not peer-accessible from here")
            for q in self.peers:
                for lo, table in ((0, self.peer_base), (hb, self.peer_fbase)):
                    hq = (C.c_ubyte * hb)(*allh[q, lo:lo + hb].tolist())
                    mapped = C.c_void_p()
                    _lib.check(self.lib.spr_p2p_open(hq, C.byref(mapped)), 'spr_p2p_open')
                    table[q] = int(mapped.value)
        except Exception as exc:                               # noqa: BLE001
            ok, why = False, f'rank {self.rank}: {exc}'
        return self._selftest(ok, why, kind)

    def _agree(self, ok, why, what):
        """-> (ok on ALL ranks, reason).  COLLECTIVE: one all-gather of every rank's verdict AND its reason, so that every rank
        reports why the first failing rank failed instead of only that it did."""
        torch = self.eng.torch
        msg = np.zeros(_REASON_BYTES, dtype=np.uint8)
        msg[0] = 1 if ok else 0
        if not ok:
            raw = np.frombuffer((why or what).encode('utf-8', errors='replace')[:_REASON_BYTES - 2], dtype=np.uint8)
            msg[1:1 + len(raw)] = raw
        allm = self.eng.to_host(self._all_gather(torch.tensor(msg, device=self.eng.device))).astype(np.uint8)
        bad = [int(q) for q in np.flatnonzero(allm[:, 0] == 0)]
        if not bad:
            return True, ''
        first = bytes(allm[bad[0], 1:]).split(b'\0')[0].decode('utf-8', errors='replace')
        if not first.startswith(f'rank {bad[0]}'):
            first = f'rank {bad[0]}: {first}'
        return False, first + (f' (ranks {bad} failed)' if len(bad) > 1 else '')

    # counter page (int64 words): arrive[b][src] | release[src] | pushed[peer] | self-test[src] | ready[0]
    def _slot(self, kind, idx, b=0):
        w = self.world
        return {'arrive': b * w + idx, 'release': 2 * w + idx, 'pushed': 3 * w + idx, 'test': 4 * w + idx, 'ready': 5 * w + idx}[kind]

    def _flag(self, page, kind, idx, b=0):
        """address of a counter in the counter page at `page`"""
        return page + 8 * self._slot(kind, idx, b)

    def _peer_flag(self, q, kind, b=0):
        """the counter peer q holds FOR ME (arrive / release / test slots are indexed by the writing rank)"""
        if self.loopback:
            # all pages are mine: the counter imaginary peer q would hold for me IS the one I await for peer q
            return self.fbase + 8 * self._slot(kind, q, b)
        return self.peer_fbase[q] + 8 * self._slot(kind, self.rank, b)

    def _selftest(self, ok, why, kind):
        """Two rounds of what a gather does, with the very calls it uses and IN THE ORDER it uses them (pushes on the copy streams
        first, the wait behind them), under a host-side watch (nothing here can block for ever: a wait that does not complete in
        SELFTEST_TIMEOUT_S is satisfied locally): every rank pushes a pattern and a counter into every peer's scratch page, joins,
        READS the page (which leaves its lines in this GPU's caches), and the second round overwrites them with another pattern --
        a rank that then still sees the first one has stale lines behind the join, and this kind of memory will not do.  The ranks
        push a little after one another, so the earlier ranks' waits are already polling when the later counters arrive.
        (The first version of this test enqueued the wait FIRST; a process maps its streams onto a few hardware queues, and
        whenever the wait's stream shared one with a copy stream the rank's own pushes sat behind its own wait until the
        time-out -- a spurious failure of the first kind of memory in 10 of 16 set-ups of a three-rank test on one GPU.)
        -> (ok on ALL ranks, reason).  COLLECTIVE."""
        torch, eng = self.eng.torch, self.eng
        t0 = time.perf_counter()
        if not self.peers:                                     # a one-rank group: nothing to exchange, nothing to test
            return ok, why
        side = torch.cuda.Stream(eng.device)
        cur = torch.cuda.current_stream(eng.device)

        def agree(flag, reason):
            return self._agree(flag, reason, f'{kind} memory: self-test failed')

        def unblock():
            self._flags[self._slot('test', 0):self._slot('test', 0) + self.world] = 1 << 40
            torch.cuda.synchronize(eng.device)

        all_ok, reason = agree(ok, why)                        # a rank that could not map its peers: nobody starts
        if not all_ok:
            return False, reason
        for rnd in (1, 2):
            value = 7 + rnd
            waiting = False
            try:
                time.sleep(0.002 * self.rank)
                pat = torch.full((4,), 1000 * rnd + self.rank, dtype=torch.int64, device=eng.device)
                ev = torch.cuda.Event()
                ev.record(cur)
                for s, q in zip(self.streams, self.peers):
                    s.wait_event(ev)
                    _lib.check(self.lib.spr_p2p_copy(self.peer_base[q] + self._scratch + 32 * self.rank, pat.data_ptr(), 32,
                                                     s.cuda_stream), 'spr_p2p_copy')
                    _lib.check(self.lib.spr_p2p_signal(self._peer_flag(q, 'test'), value, s.cuda_stream), 'spr_p2p_signal')
                tab = _ptr_array([self._flag(self.fbase, 'test', q) for q in self.peers])
                with torch.cuda.stream(side):
                    _lib.check(self.lib.spr_p2p_flags_wait(tab, len(self.peers), value, self.SELFTEST_TIMEOUT_S + 5.0,
                                                           None, side.cuda_stream), 'spr_p2p_flags_wait')
                waiting = True
                deadline = time.perf_counter() + self.SELFTEST_TIMEOUT_S
                while not side.query():
                    if time.perf_counter() > deadline:
                        ok, why = False, (f'{kind} memory: a wait on a counter written by a peer did not complete in '
                                          f'{self.SELFTEST_TIMEOUT_S:.0f} s')
                        unblock()
                        break
                    time.sleep(0.0005)
                torch.cuda.synchronize(eng.device)
                if ok:
                    cur.wait_stream(side)
                    page = self._mem[self._scratch:self._scratch + _SCRATCH_BYTES].view(torch.int64)
                    got = (page + 0).cpu().numpy().reshape(-1, 4)          # read by a KERNEL (the add), like a consumer would
                    rows = [q for q in self.peers if not np.array_equal(got[q], [1000 * rnd + q] * 4)]
                    if rows:
                        ok, why = False, (f'{kind} memory: round {rnd}, the pattern pushed by ranks {rows} is not what a '
                                          'kernel reads behind the join' + (' (stale lines)' if rnd == 2 else ''))
            except Exception as exc:                           # noqa: BLE001 -- any failure means "not available"
                ok, why = False, f'self-test: {exc}'
                if waiting:
                    try:
                        unblock()
                    except Exception:                          # noqa: BLE001
                        pass
            all_ok, reason = agree(ok, why)
            if not all_ok:
                return False, reason
        self.selftest_report = dict(seconds=time.perf_counter() - t0, memory=kind)
        return True, ''

    # ------------------------------------------------------------------ one gather
    def _table(self, key, make):
        t = self._tabs.get(key)
        if t is None:
            t = self._tabs[key] = make()
        return t

    def release_timeout_s(self):
        return float(self.RELEASE_TIMEOUT_S) if self.RELEASE_TIMEOUT_S else min(3600.0, 3.0 * float(self.JOIN_TIMEOUT_S))

    def begin(self):
        """Enter gather k (COLLECTIVE): tell the peers which of my copies they may overwrite, and hand out the tensor the
        reconstruct kernel writes this rank's block into -- the (n_p, n_total) view of buffer k % n_buf."""
        eng, torch = self.eng, self.eng.torch
        t_host = time.perf_counter()
        self.check()
        st = torch.cuda.current_stream(eng.device).cuda_stream
        if self.peers:
            tab = self._table('release', lambda: _ptr_array([self._peer_flag(q, 'release') for q in self.peers]))
            _lib.check(self.lib.spr_field_gather_p2p_release(tab, len(self.peers), self.k, st), 'spr_field_gather_p2p_release')
        b = self.k % self.n_buf
        n_p, n_total = self.shape
        self.host_ms['begin'] += 1e3 * (time.perf_counter() - t_host)
        return self._mem[b * self.field_bytes:b * self.field_bytes + n_p * n_total * 8].view(torch.float64).view(n_p, n_total)

    def push(self, first, n_loc):
        """My block -- columns [first, first + n_loc) of the tensor begin() returned, written by work already enqueued on the
        current stream -- into every peer's copy.  Returns at once; the copies run on the copy streams."""
        eng, torch = self.eng, self.eng.torch
        t_host = time.perf_counter()
        b = self.k % self.n_buf
        n_p, n_total = self.shape
        if self.peers:
            ready = 0
            if self._order_by_counter:
                # "my block is written": ONE single-wave kernel on the compute stream raises ready = k + 1 in my counter page (a
                # system-scope release behind the reconstruct kernel); the copy streams' wait kernels poll it next to the peers'
                # releases.  No event of the compute stream: recording one and making three copy streams wait for it cost 0.25 ms of
                # host time per push while the compute stream was busy (profiles/r06_p2p_push_host.txt)
                ready = self._flag(self.fbase, 'ready', 0)
                tab = self._table('ready', lambda: _ptr_array([ready]))
                _lib.check(self.lib.spr_p2p_flags_set(tab, 1, self.k + 1, torch.cuda.current_stream(eng.device).cuda_stream),
                           'spr_p2p_flags_set')
            else:
                # (a fresh event per push: re-recording ONE event while its previous record is still awaited by the copy streams
                #  cost 0.5-1.0 ms of host time per push in the step loop)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(eng.device))
                for s in self._pool:
                    s.wait_event(ev)
            t_call = time.perf_counter()
            self.host_ms['push_order'] += 1e3 * (t_call - t_host)
            field = self.base + b * self.field_bytes
            release_value = max(self.k - self.n_buf + 1, 0)
            tabs = self._table(('push', b), lambda: (
                _ptr_array([self.peer_base[q] + b * self.field_bytes for q in self.peers]),
                _ptr_array([self._flag(self.fbase, 'release', q) for q in self.peers]),
                _ptr_array([self._peer_flag(q, 'arrive', b) for q in self.peers]),
                _ptr_array([self._flag(self.fbase, 'pushed', q) for q in self.peers]),
                _ptr_array([s.cuda_stream for s in self.streams])))
            _lib.check(self.lib.spr_field_gather_p2p(
                field, n_total, n_p, int(first), int(n_loc), len(self.peers), tabs[0], tabs[1], release_value,
                self.release_timeout_s(), tabs[2], self.k + 1, tabs[3], tabs[4], self._status.data_ptr() + 16, ready, self.k + 1),
                'spr_field_gather_p2p')
            self.host_ms['push_call'] += 1e3 * (time.perf_counter() - t_call)
        joined_k = self.k
        self.k += 1
        self.host_ms['push'] += 1e3 * (time.perf_counter() - t_host)
        self.host_ms['calls'] += 1
        return joined_k

    def join(self, k):
        """The current stream waits (one kernel) for the peers' blocks of gather k and for my own pushes to have left."""
        eng, torch = self.eng, self.eng.torch
        t_host = time.perf_counter()
        b = k % self.n_buf
        if self.peers:
            tab = self._table(('join', b), lambda: _ptr_array([self._flag(self.fbase, 'arrive', q, b) for q in self.peers]
                                                              + [self._flag(self.fbase, 'pushed', q) for q in self.peers]))
            _lib.check(self.lib.spr_field_gather_p2p_join(tab, 2 * len(self.peers), k + 1, self.JOIN_TIMEOUT_S,
                                                          self._status.data_ptr(),
                                                          torch.cuda.current_stream(eng.device).cuda_stream),
                       'spr_field_gather_p2p_join')
        self.host_ms['join'] += 1e3 * (time.perf_counter() - t_host)

    def check(self):
        """Did a wait give up?  A read of four words of page-locked host memory the kernels write on their way out: free, so
        begin() calls it at every gather -- a dead peer surfaces as a RuntimeError naming the missing counter at the latest one
        gather after the join that waited for it, instead of as a stale field.  Three causes, all RuntimeError:
          * the join timed out on arrive[q] (the block of rank q never came) or pushed[q] (my own push to q never left);
          * the join read a POISONED counter: the rank that raised it had given up waiting for a release and pushed anyway
            (arrive[q]: rank q did, into MY copy while I had not released it -- what I was handed before may have been
            overwritten under my readers; pushed[q]: I did, into rank q's);
          * a push of mine gave up waiting for release[q] (words 2-3): rank q had not entered the gather within
            RELEASE_TIMEOUT_S -- the copy went ahead (it cannot be taken back) with the poison behind it."""
        st = self._status_np
        if not (st[0] or st[2]):
            return
        n = len(self.peers)
        msgs = []
        if st[2] and int(st[2]) - 1 >= n:
            msgs.append(f'a push of rank {self.rank} gave up waiting for its own counter ready[0] (the kernel that writes the block; at '
                        f'{int(st[3]) & ~self._poison}) after {self.release_timeout_s():.0f} s; the arrival counters behind it carry '
                        'the poison bit')
        elif st[2]:
            q = self.peers[min(int(st[2]) - 1, n - 1)]
            seen = int(st[3])
            msgs.append(f'a push of rank {self.rank} gave up waiting for counter release[{q}] (rank {q} letting go of its copy '
                        f'of the field; at {seen & ~self._poison}{", poisoned" if seen & self._poison else ""}) after '
                        f'{self.release_timeout_s():.0f} s: the block was copied all the same (an SDMA command cannot be taken '
                        f'back) and the arrival counters behind it were raised with the poison bit -- the join of this gather '
                        f'fails on rank {q} and here')
        if st[0]:
            i = int(st[0]) - 1
            seen = int(st[1])
            q = self.peers[i] if i < n else self.peers[min(i - n, n - 1)]
            name = f'arrive[{q}]' if i < n else f'pushed[{q}]'
            if seen & self._poison:
                what = (f'rank {q} pushed its block without my release (it had given up waiting for it): the field I was reading '
                        'may have been overwritten' if i < n else
                        f'my own push to rank {q} went ahead without its release')
                msgs.append(f'the join of rank {self.rank} read a POISONED counter {name} ({what})')
            else:
                what = f'the block of rank {q}' if i < n else f'my own push to rank {q}'
                msgs.append(f'rank {self.rank} gave up waiting for counter {name} ({what}) after {self.JOIN_TIMEOUT_S:.0f} s '
                            f'(counter at {seen})')
        k = self.k - 1
        st[:] = 0
        raise RuntimeError('p2p field exchange, gather ' + str(k) + ': ' + '; '.join(msgs) + '; the field handed out by that '
                           'join is incomplete and this exchange object must not be used again')

    def arrived(self, k):
        """Host-side look at the arrival counters of gather k (one small D2H copy): which peers' blocks are still missing."""
        b = k % self.n_buf
        fl = self._flags[b * self.world:(b + 1) * self.world].cpu().numpy()
        return [q for q in self.peers if fl[q] < k + 1]

    def abandon(self):
        """After a failed exchange: raise every counter of MY page far beyond any gather count AND poison it, so that the waits my
        copy streams and my join kernels still hold (all of them poll my page) drain at once instead of sitting in the queues
        until the process ends -- as FAILED waits: whatever my copy streams still push carries the poison to the peers.  The
        buffers stay allocated and mapped -- peers may still be writing -- and the object is not used again."""
        torch = self.eng.torch
        if self._flags is not None:
            side = torch.cuda.Stream(self.eng.device)          # not behind a join kernel that is still polling
            with torch.cuda.stream(side):
                self._flags.fill_((1 << 40) | self._poison)
            deadline = time.perf_counter() + 2.0               # bounded: the fill may share a hardware queue with what it is to release
            while not side.query() and time.perf_counter() < deadline:
                time.sleep(0.001)
        self._status_np[:] = 0

    # ------------------------------------------------------------------ teardown
    def close(self, collective=True, rendezvous=None):
        """Unmap the peers' buffers and free mine.  With ``collective`` every rank first drains its streams and the ranks
        meet (an all-gather of one number) between unmapping and freeing, so nobody frees what a peer still has mapped.
        ``rendezvous``: whether the ranks meet -- None: when this rank holds buffers (a user's close() of a set-up exchange:
        every rank does); True / False: decided by the caller from what ALL ranks know (ensure(): a rank whose own allocation
        failed must still meet the others)."""
        have = self.base is not None or self.fbase is not None
        meet = collective and self.world > 1 and not self.loopback and (have if rendezvous is None else bool(rendezvous))
        if not have and not meet:
            return
        torch = self.eng.torch
        if have:
            torch.cuda.synchronize(self.eng.device)
        if not self.loopback:
            for table in (self.peer_base, self.peer_fbase):
                for q, p in list(table.items()):
                    try:
                        _lib.check(self.lib.spr_p2p_close(p), 'spr_p2p_close')
                    except Exception:                          # noqa: BLE001 -- teardown
                        pass
        self.peer_base, self.peer_fbase = {}, {}
        self._tabs = {}
        if meet:
            try:
                self._all_gather(torch.zeros(1, device=self.eng.device))
            except Exception:                                  # noqa: BLE001 -- the group may be gone at interpreter exit
                pass
        self._mem = self._flags = None
        for attr in ('base', 'fbase'):
            p = getattr(self, attr)
            setattr(self, attr, None)
            if p is not None:
                try:
                    _lib.check(self.lib.spr_p2p_free(p), 'spr_p2p_free')
                except Exception:                              # noqa: BLE001 -- teardown
                    pass
        self.field_bytes = 0
