"""Host plumbing of the CU-free field gather (include/spr_hip.h: spr_p2p_*, spr_field_gather_p2p*; csrc/p2p.hip).

``reconstruct()`` of a row-sharded SPR ends with every rank's block of the field on every rank (reference: the (n, n_p)
array of sparse_sensing.py:371-375 is whole on every caller).  Over RCCL that is an all-gather whose device kernel cannot
share a compute unit with the Gram / projection workgroups; here every rank keeps ONE persistent copy of the field in a
buffer the other ranks of the node have mapped (interprocess handles, exchanged once through torch.distributed), the
reconstruct kernel writes the rank's own block straight into it, and the block is pushed into the same place of every peer's
copy by the SDMA engines -- one stream per peer, a 64-bit arrival counter written behind the copies, awaited by the
consumer's stream.  No compute unit is used, so the exchange really runs under the next fit()'s MFMA-bound Gram pass.

PyTorch is plumbing: streams, the device context, and ``torch.distributed`` as the channel for the 64 handle bytes per
rank.  Buffers come from the library (an interprocess handle needs the base pointer of an allocation).

Protocol (counters only ever grow; k = number of gathers issued so far on this object, identical on all ranks because a
gather is a collective call; b = k % n_buf, n_buf = 1 unless asked otherwise):
  release   on entering gather k a rank raises release[me] = k in every peer's flag page, on its COMPUTE stream -- behind every
            kernel that read an earlier field: "the fields I was handed before gather k are dead";
  push      per peer p, on that peer's copy stream: wait until release[p] >= k - n_buf + 1 (in MY flag page: p no longer reads
            what its buffer b holds), copy my block into p's buffer b, raise arrive[b][me] = k + 1 in p's flag page;
  join      the consumer's stream waits until arrive[b][p] >= k + 1 for every peer p (MY flag page) and for my own pushes to
            have left (their source is my buffer b).
The field handed out is a VIEW of buffer b: it stays valid until this rank enters its next gather.  (A second buffer only lets
the peers' COPY streams run a step ahead of a slow rank; no compute stream ever waits for a push, so one buffer is the default.)
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib

_FLAG_BYTES = 4096
_SCRATCH_BYTES = 4096
_MAX_WORLD = 120                     # 4 counter arrays of `world` uint64 in the 4 KB flag page


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
    """One per sharded ROM object and field shape; every method that says COLLECTIVE must be called by all ranks."""

    SELFTEST_TIMEOUT_S = 10.0

    def __init__(self, eng, world, rank, all_gather, double_buffer=False):
        if world > _MAX_WORLD:
            raise P2PUnavailable(f'{world} ranks exceed the {_MAX_WORLD} the flag page holds')
        self.eng, self.world, self.rank = eng, int(world), int(rank)
        self.lib = eng.lib
        self._all_gather = all_gather                          # tensor -> (world, *shape) tensor, COLLECTIVE
        self.n_buf = 2 if double_buffer else 1
        self.k = 0                                             # gathers issued
        self.base = None                                       # my buffer (device pointer)
        self.peer_base = {}                                    # rank -> mapped pointer
        self.shape = None                                      # (n_p, n_total)
        self.field_bytes = 0
        torch = eng.torch
        self.peers = [q for q in range(self.world) if q != self.rank]
        self.streams = [torch.cuda.Stream(eng.device) for _ in self.peers]
        self._pushed = None                                    # events behind the pushes of the last gather
        self.selftest_report = None

    # ------------------------------------------------------------------ set-up (COLLECTIVE)
    def ensure(self, n_p, n_total):
        """Buffers for an (n_p, n_total) float64 field; (re)allocated and exchanged when the shape grows.  COLLECTIVE when
        it allocates -- every rank sees the same shapes, so every rank takes the same branch."""
        n_p, n_total = int(n_p), int(n_total)
        need = -(-n_p * n_total * 8 // 4096) * 4096
        if self.base is not None and need <= self.field_bytes:
            self.shape = (n_p, n_total)
            return
        self.close()
        torch = self.eng.torch
        self.field_bytes = need
        total = self.n_buf * need + _SCRATCH_BYTES + _FLAG_BYTES
        hb = int(self.lib.spr_p2p_handle_bytes())
        handle = (C.c_ubyte * hb)()
        ok, why = True, ''
        try:
            base = C.c_void_p()
            _lib.check(self.lib.spr_p2p_alloc(total, C.byref(base), handle), 'spr_p2p_alloc')
            self.base, self.total_bytes = int(base.value), total
            self._mem = torch.as_tensor(_Raw(self.base, total), device=self.eng.device)  # uint8 view of my buffer
            self._mem[self.n_buf * need:].zero_()              # scratch + flags
            torch.cuda.synchronize(self.eng.device)
        except Exception as exc:                               # noqa: BLE001 -- "not available", decided together below
            ok, why = False, f'rank {self.rank}: {exc}'
        # every rank's handle bytes and whether it has a buffer at all: one exchange, so that all ranks go on or give up together
        mine = np.zeros(hb + 8, dtype=np.uint8)
        mine[:hb] = np.frombuffer(bytes(handle), dtype=np.uint8)
        mine[hb] = 1 if ok else 0
        allh = self.eng.to_host(self._all_gather(torch.tensor(mine, device=self.eng.device)))      # (world, hb + 8)
        if not allh[:, hb].all():
            bad = [int(q) for q in np.flatnonzero(allh[:, hb] == 0)]
            self.close(collective=False)
            raise P2PUnavailable(why or f'ranks {bad} could not allocate an exportable buffer')
        try:
            for q in self.peers:
                hq = (C.c_ubyte * hb)(*allh[q, :hb].tolist())
                mapped = C.c_void_p()
                _lib.check(self.lib.spr_p2p_open(hq, C.byref(mapped)), 'spr_p2p_open')
                self.peer_base[q] = int(mapped.value)
        except Exception as exc:                               # noqa: BLE001
            ok, why = False, f'rank {self.rank}: {exc}'
        self.shape = (n_p, n_total)
        self.k = 0
        ok, why = self._selftest(ok, why)
        if not ok:
            self.close(collective=False)
            raise P2PUnavailable(why)

    def _flag(self, base, kind, idx, b=0):
        """address of a counter in the flag page of the buffer at `base`: kind 0 arrive[b][idx], 1 release[idx], 2 self-test[idx]"""
        off = self.n_buf * self.field_bytes + _SCRATCH_BYTES
        slot = {0: b * self.world + idx, 1: 2 * self.world + idx, 2: 3 * self.world + idx}[kind]
        return base + off + 8 * slot

    def _my_flags(self):
        off = self.n_buf * self.field_bytes + _SCRATCH_BYTES
        return self._mem[off:off + _FLAG_BYTES].view(self.eng.torch.int64)

    def _selftest(self, ok, why):
        """Every rank pushes a 4 KB pattern and a counter into every peer's buffer and waits for the peers' with the very
        calls the gather uses; nothing here can block for ever: the stream wait is watched from the host and, on a timeout,
        satisfied locally.  -> (ok on ALL ranks, reason).  COLLECTIVE."""
        torch, eng = self.eng.torch, self.eng
        t0 = time.perf_counter()
        report = {}
        scratch_off = self.n_buf * self.field_bytes
        if ok:
            try:
                pat = torch.full((_SCRATCH_BYTES // 8,), 1000 + self.rank, dtype=torch.int64, device=eng.device)
                side = torch.cuda.Stream(eng.device)
                with torch.cuda.stream(side):                  # the waits first: they must see values that arrive later
                    for q in self.peers:
                        _lib.check(self.lib.spr_p2p_wait(self._flag(self.base, 2, q), 7, side.cuda_stream), 'spr_p2p_wait')
            except Exception as exc:                           # noqa: BLE001 -- any failure means "not available"
                ok, why = False, f'self-test set-up: {exc}'
        # everybody has its waits enqueued (or has failed) before anybody writes
        oks = self.eng.to_host(self._all_gather(torch.tensor([1.0 if ok else 0.0], device=eng.device)))
        if not oks.all():
            bad = [int(q) for q in np.flatnonzero(oks.reshape(-1) == 0)]
            if ok:                                             # my waits are enqueued: satisfy them myself
                self._my_flags()[3 * self.world:4 * self.world] = 7
                torch.cuda.synchronize(eng.device)
            return False, why or f'ranks {bad} could not map their peers'
        try:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(eng.device))
            for s, q in zip(self.streams, self.peers):
                s.wait_event(ev)
                # pattern into MY 32-byte-per-rank slot... the scratch page of peer q, at offset 32 * rank
                _lib.check(self.lib.spr_p2p_copy(self.peer_base[q] + scratch_off + 32 * self.rank, pat.data_ptr(), 32,
                                                 s.cuda_stream), 'spr_p2p_copy')
                _lib.check(self.lib.spr_p2p_signal(self._flag(self.peer_base[q], 2, self.rank), 7, s.cuda_stream),
                           'spr_p2p_signal')
            deadline = time.perf_counter() + self.SELFTEST_TIMEOUT_S
            while not side.query():
                if time.perf_counter() > deadline:
                    ok, why = False, 'a stream wait on a counter written by a peer did not complete in ' \
                                     f'{self.SELFTEST_TIMEOUT_S:.0f} s'
                    self._my_flags()[3 * self.world:4 * self.world] = 7      # unblock the side stream
                    break
                time.sleep(0.0005)
            torch.cuda.synchronize(eng.device)
            if ok:
                got = self._mem[scratch_off:scratch_off + _SCRATCH_BYTES].view(torch.int64).cpu().numpy().reshape(-1, 4)
                want = np.array([[1000 + q] * 4 for q in range(self.world)])
                rows = [q for q in self.peers if not np.array_equal(got[q], want[q])]
                if rows:
                    ok, why = False, f'the pattern pushed by ranks {rows} did not arrive intact'
        except Exception as exc:                               # noqa: BLE001
            ok, why = False, f'self-test: {exc}'
            try:
                self._my_flags()[3 * self.world:4 * self.world] = 7
                torch.cuda.synchronize(eng.device)
            except Exception:                                  # noqa: BLE001
                pass
        oks = self.eng.to_host(self._all_gather(torch.tensor([1.0 if ok else 0.0], device=eng.device)))
        report['seconds'] = time.perf_counter() - t0
        self.selftest_report = report
        if not oks.all():
            bad = [int(q) for q in np.flatnonzero(oks.reshape(-1) == 0)]
            return False, why or f'self-test failed on ranks {bad}'
        return True, ''

    # ------------------------------------------------------------------ one gather
    def begin(self):
        """Enter gather k (COLLECTIVE): tell the peers which of my copies they may overwrite, and hand out the tensor the
        reconstruct kernel writes this rank's block into -- the (n_p, n_total) view of buffer k % n_buf."""
        eng, torch = self.eng, self.eng.torch
        cur = torch.cuda.current_stream(eng.device)
        st = cur.cuda_stream
        if self._pushed:                                       # a gather nobody joined: its pushes still read my copy
            for e in self._pushed:
                cur.wait_event(e)
            self._pushed = None
        if self.peers:
            tab = _ptr_array([self._flag(self.peer_base[q], 1, self.rank) for q in self.peers])
            _lib.check(self.lib.spr_field_gather_p2p_release(tab, len(self.peers), self.k, st), 'spr_field_gather_p2p_release')
        b = self.k % self.n_buf
        n_p, n_total = self.shape
        return self._mem[b * self.field_bytes:b * self.field_bytes + n_p * n_total * 8].view(torch.float64).view(n_p, n_total)

    def push(self, first, n_loc):
        """My block -- columns [first, first + n_loc) of the tensor begin() returned, written by work already enqueued on the
        current stream -- into every peer's copy.  Returns at once; the copies run on the copy streams."""
        eng, torch = self.eng, self.eng.torch
        b = self.k % self.n_buf
        n_p, n_total = self.shape
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(eng.device))
        for s in self.streams:
            s.wait_event(ev)
        if self.peers:
            field = self.base + b * self.field_bytes
            release_value = max(self.k - self.n_buf + 1, 0)
            _lib.check(self.lib.spr_field_gather_p2p(
                field, n_total, n_p, int(first), int(n_loc), len(self.peers),
                _ptr_array([self.peer_base[q] + b * self.field_bytes for q in self.peers]),
                _ptr_array([self._flag(self.base, 1, q) for q in self.peers]), release_value,
                _ptr_array([self._flag(self.peer_base[q], 0, self.rank, b) for q in self.peers]), self.k + 1,
                _ptr_array([s.cuda_stream for s in self.streams])), 'spr_field_gather_p2p')
        self._pushed = []
        for s in self.streams:
            e = torch.cuda.Event()
            e.record(s)
            self._pushed.append(e)
        joined_k = self.k
        self.k += 1
        return joined_k

    def join(self, k):
        """The current stream waits for the peers' blocks of gather k and for my own pushes to have left."""
        eng, torch = self.eng, self.eng.torch
        cur = torch.cuda.current_stream(eng.device)
        b = k % self.n_buf
        if self.peers:
            tab = _ptr_array([self._flag(self.base, 0, q, b) for q in self.peers])
            _lib.check(self.lib.spr_field_gather_p2p_join(tab, len(self.peers), k + 1, cur.cuda_stream),
                       'spr_field_gather_p2p_join')
        if k == self.k - 1 and self._pushed:
            for e in self._pushed:
                cur.wait_event(e)
            self._pushed = None

    def arrived(self, k):
        """Host-side look at the arrival counters of gather k (one small D2H copy): which peers' blocks are still missing."""
        b = k % self.n_buf
        fl = self._my_flags()[b * self.world:(b + 1) * self.world].cpu().numpy()
        return [q for q in self.peers if fl[q] < k + 1]

    # ------------------------------------------------------------------ teardown
    def close(self, collective=True):
        """Unmap the peers' buffers and free mine.  With ``collective`` every rank first drains its streams and the ranks
        meet (an all-gather of one number) between unmapping and freeing, so nobody frees what a peer still has mapped."""
        if self.base is None:
            return
        torch = self.eng.torch
        torch.cuda.synchronize(self.eng.device)
        for q, p in list(self.peer_base.items()):
            try:
                _lib.check(self.lib.spr_p2p_close(p), 'spr_p2p_close')
            except Exception:                                  # noqa: BLE001 -- teardown
                pass
        self.peer_base = {}
        if collective and self.world > 1:
            try:
                self._all_gather(torch.zeros(1, device=self.eng.device))
            except Exception:                                  # noqa: BLE001 -- the group may be gone at interpreter exit
                pass
        self._mem = None
        try:
            _lib.check(self.lib.spr_p2p_free(self.base), 'spr_p2p_free')
        finally:
            self.base = None
            self.field_bytes = 0
