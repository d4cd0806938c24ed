"""HipEngine: the device half of the SPR path -- torch tensors in, libspr_hip.so calls out.

PyTorch is plumbing here: it owns the HBM allocations (``torch.empty(..., device='cuda')``),
the current HIP stream and, through ``torch.distributed``, the RCCL communicator.  All
arithmetic on the snapshot matrix and the basis happens in the hand-written gfx950 kernels
reached through the C ABI (``include/spr_hip.h``).  Every method enqueues on the current
stream and returns device tensors; nothing here synchronises the host.

The ROM/SPR classes talk to an *engine* object with exactly this method set.  The product
has one engine (this one).  ``tests/numpy_engine.py`` holds a NumPy stand-in with the same
interface that exists only so the sharding/collective logic can be exercised with the gloo
backend on machines without a GPU; nothing in this package imports it.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _ptr(t):
    return t.data_ptr() if t is not None else None


_NP_OF = {'torch.float64': np.float64, 'torch.float32': np.float32, 'torch.int64': np.int64, 'torch.int32': np.int32,
          'torch.uint8': np.uint8, 'torch.int8': np.int8, 'torch.int16': np.int16, 'torch.bool': np.bool_}


class HipEngine:
    name = 'hip-gfx950'

    supports_row_norms = True      # project(norms=...) / qr_begin(norms=...): see ROM.placement_norms
    supports_gram_out = True       # stats_gram(gram_out=, fstats_out=): results written into the caller's buffers

    def __init__(self, device=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError('openmeasure_amd needs an AMD GPU visible to PyTorch-ROCm; '
                               'there is no CPU fallback for the SPR kernels.')
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self._ws = {}
        self._timing = {}
        import os
        self._force_stream = os.environ.get('SPR_PROJECT_STREAM') == '1'   # A/B runs: streamed-W projection for every shape
        self._dl_kernel = os.environ.get('SPR_DL_KERNEL', '1') != '0'      # A/B: small downloads by kernel + polled ticket (default) or by copy + event
        self._stage = None                                   # ring of pinned host staging buffers for small uploads
        # page-locked buffers, the copy threads and the side stream go back while the interpreter and the HIP runtime are
        # still whole (atexit runs before module teardown): what is left to destructors at process exit runs in no
        # particular order against the runtime's own shutdown
        import atexit
        import weakref
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close(wait=False))

    def close(self, wait=True):
        """Give back what the engine holds outside PyTorch's device allocator: the page-locked staging buffers, the copy
        threads of the staged downloads.  Idempotent; the engine can be used again afterwards (buffers are
        made on demand).  Host arrays handed out earlier (page-locked results) stay valid: they own their memory.
        ``wait=False`` (the atexit hook): without the device synchronisation in front -- a process on its way out must not
        block on a stream that waits for a peer which is gone (the copy streams of the p2p exchange wait for counters)."""
        torch = self.torch
        if wait:
            try:
                torch.cuda.synchronize(self.device)
            except RuntimeError:
                pass
        pool = self.__dict__.pop('_copy_pool', None)
        if pool is not None:
            pool.shutdown(wait=wait, cancel_futures=not wait)
        self._stage = None
        for k in ('_dstage2', '_dstage_slots', '_dl', '_reuse', '_side'):
            self.__dict__.pop(k, None)
        try:
            torch._C._host_emptyCache()                       # cached page-locked blocks back to the OS
        except (AttributeError, RuntimeError):
            pass

    # ---- plumbing ---------------------------------------------------------------------
    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def empty(self, shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.torch.float64, device=self.device)

    def zeros(self, shape, dtype=None):
        return self.torch.zeros(shape, dtype=dtype or self.torch.float64, device=self.device)

    _STAGE_BYTES = 1 << 20
    _STAGE_SLOTS = 8

    def to_device(self, a, dtype=None):
        """Host ndarray -> contiguous device tensor (float64 unless told otherwise).  Small arrays are copied
        into one of a ring of pinned staging buffers and carried to the device by a kernel on the current stream
        (spr_upload_bytes): the host never blocks, and no copy-engine dependency sits in front of the next
        kernel -- with hipMemcpyAsync there, the projection was measured to start 10/20/30 ms late in every
        other fit() at config 3 (tools/fit_probe.py).  A slot is reused only after the event recorded behind
        its upload has completed."""
        torch = self.torch
        if dtype is None:
            dtype = torch.float64
        t = torch.as_tensor(np.ascontiguousarray(a))
        if t.dtype != dtype:
            t = t.to(dtype)
        nbytes = t.numel() * t.element_size()
        if nbytes == 0 or nbytes > self._STAGE_BYTES or nbytes % 8:
            return t.to(device=self.device).contiguous()
        if self._stage is None:
            self._stage = [[torch.empty(self._STAGE_BYTES, dtype=torch.uint8, pin_memory=True), None]
                           for _ in range(self._STAGE_SLOTS)]
            self._stage_next = 0
        slot = self._stage[self._stage_next]
        self._stage_next = (self._stage_next + 1) % self._STAGE_SLOTS
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:nbytes].view(dtype).view(t.shape).copy_(t)
        out = torch.empty(t.shape, dtype=dtype, device=self.device)
        _lib.check(self.lib.spr_upload_bytes(out.data_ptr(), slot[0].data_ptr(), nbytes, self._stream()),
                   'spr_upload_bytes')
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.device))
        return out

    _HOST_STAGE_BYTES = 64 << 20
    #: to_host(result=True) -- what is handed to the CALLER: fields, the basis -- above this size comes back in a page-locked
    #: tensor of its own (_to_host_big: one DMA at the PCIe rate, the ndarray handed out IS that memory, counted against the
    #: budget of _pinned_result); below it, and for every internal download up to 64 MiB, through the shared pinned stage and a
    #: host copy -- 1.8 of the 2.7 ms a 32 MB field took to reach the caller at config 2 were that copy into fresh pages.
    #: Internal downloads stay on the stage as before: they are not handed out, and a page-locked block PyTorch's host allocator
    #: has just taken back is only reusable once the stream that copied into it has passed the point of its release -- a hot path
    #: that took a new block per call would keep pinning fresh memory while the stream is busy.  (The downloads inside a step
    #: are all below 1 MiB -- the combined Gram matrix is m x m -- and go through spr_download_bytes anyway.)
    _PINNED_MIN_BYTES = 4 << 20

    def to_host(self, t, then=None, result=False):
        """Device tensor -> fresh host ndarray.  ``then``: called after the copy has been ENQUEUED and before the host
        blocks on it -- work it launches queues up behind the copy (fit()'s gap filler).  ``result``: the array goes to the caller
        (see _PINNED_MIN_BYTES).  Downloads up to 64 MiB (4 MiB for results: statistics, Gram blocks, flags, Theta,
        coefficient vectors) come back through a pinned buffer: a D2H copy into pageable memory in the middle of
        fit() left the compute queue stalled for 10/20/30 ms in every other call at config 3 (tools/fit_probe.py:
        gap between the Gram and projection kernels 3.9 ms with the pinned target, 4-37 ms without)."""
        torch = self.torch
        t = t.detach()
        nbytes = t.numel() * t.element_size()
        if (0 < nbytes <= self._DL_KERNEL_BYTES and nbytes % 8 == 0 and t.is_cuda and t.is_contiguous() and t.data_ptr() % 8 == 0
                and str(t.dtype) in _NP_OF and self._dl_kernel):
            return self._to_host_small(t, nbytes, then)
        if nbytes == 0 or nbytes > (self._PINNED_MIN_BYTES if result else self._HOST_STAGE_BYTES) or not t.is_cuda:
            out = self._to_host_big(t) if (t.is_cuda and nbytes) else t.cpu().numpy()
            if then is not None:
                then()
            return out
        # one landing buffer PER NESTING DEPTH: `then` may itself download (a deferred reconstruct launched from fit()'s host
        # gap runs collective set-up with its own to_host() calls) -- a nested call must not land in the buffer, or behind the
        # event, this call is about to read
        depth = self._dl_depth
        slots = self.__dict__.setdefault('_dstage_slots', {})
        ent = slots.get(depth)
        if ent is None or ent[0].numel() < nbytes:
            slots[depth] = None
            ent = slots[depth] = (torch.empty(max(1 << 20, -(-nbytes // (1 << 20)) << 20), dtype=torch.uint8, pin_memory=True),
                                  torch.cuda.Event())
        stage, ev = ent
        buf = stage[:nbytes].view(t.dtype).view(t.shape)
        buf.copy_(t, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        if then is not None:
            self._dl_depth = depth + 1
            try:
                then()
            finally:
                self._dl_depth = depth
        ev.synchronize()
        return buf.numpy().copy()

    _dl_depth = 0                       # how many to_host() calls are inside their `then` hook right now

    _DL_KERNEL_BYTES = 1 << 20          # downloads up to this size go through spr_download_bytes
    _DL_SPIN_S = 2e-3                   # the host polls the ticket this long before it blocks on the event instead

    def _to_host_small(self, t, nbytes, then):
        """Small results (the m x m Gram matrix with its statistics, flags, candidate records): a kernel writes them into a
        page-locked buffer and stores a ticket behind them (spr_download_bytes); the host polls the ticket in its own memory.
        No copy-engine launch, no wake-up of a blocked thread: 25-40 us less in the host gap of fit() than the event-based
        path (config 2: a gap of 0.3 ms in a 1.5 ms step).  Waits longer than _DL_SPIN_S fall back to blocking on an event
        (a 93 ms Gram pass is not worth a spinning core)."""
        import time
        torch = self.torch
        # one buffer + ticket PER NESTING DEPTH (see to_host): a download issued from inside `then` takes the next slot, so the
        # payload and the ticket this call polls are its own
        depth = self._dl_depth
        slots = self.__dict__.setdefault('_dl', {})
        dl = slots.get(depth)
        if dl is None:
            buf = torch.empty(self._DL_KERNEL_BYTES + 64, dtype=torch.uint8, pin_memory=True)
            arr = buf.numpy()
            dl = slots[depth] = dict(buf=buf, data=arr[:self._DL_KERNEL_BYTES],
                                     ticket=arr[self._DL_KERNEL_BYTES:self._DL_KERNEL_BYTES + 8].view(np.uint64), seq=0,
                                     ev=torch.cuda.Event())
            dl['ticket'][0] = 0
        dl['seq'] += 1
        seq = dl['seq']
        st = torch.cuda.current_stream(self.device)
        _lib.check(self.lib.spr_download_bytes(dl['buf'].data_ptr(), t.data_ptr(), nbytes, dl['buf'].data_ptr() + self._DL_KERNEL_BYTES,
                                               seq, st.cuda_stream), 'spr_download_bytes')
        dl['ev'].record(st)
        if then is not None:
            self._dl_depth = depth + 1
            try:
                then()
            finally:
                self._dl_depth = depth
        ticket = dl['ticket']
        t_end = time.perf_counter() + self._DL_SPIN_S
        while ticket[0] != seq:
            if time.perf_counter() > t_end:
                dl['ev'].synchronize()
                break
        out = np.empty(tuple(t.shape), dtype=dl['data'][:nbytes].view(_NP_OF[str(t.dtype)]).dtype)
        np.copyto(out.reshape(-1), dl['data'][:nbytes].view(out.dtype))
        return out

    def upload_reuse(self, key, a):
        """Host float64 ndarray -> a device tensor that is OVERWRITTEN by the next call with the same key and shape (one
        page-locked staging buffer and one device buffer per key).  For an operand whose previous consumer is known to have
        finished -- W of fit(): its only reader is the projection of the previous fit(), which lies in front of the Gram
        download the host has just waited for.  A third of the host time of to_device(): no allocation, no event."""
        torch = self.torch
        a = np.ascontiguousarray(a, dtype=np.float64)
        ent = self.__dict__.setdefault('_reuse', {}).get(key)
        if not self._dl_kernel:
            return self.to_device(a)
        if ent is None or ent[2].shape != a.shape:
            pin = torch.empty(max(a.nbytes, 8), dtype=torch.uint8, pin_memory=True)
            ent = self._reuse[key] = (pin, pin.numpy()[:a.nbytes].view(np.float64).reshape(a.shape), torch.empty(a.shape, dtype=torch.float64,
                                                                                                              device=self.device))
        np.copyto(ent[1], a)
        if a.nbytes:
            _lib.check(self.lib.spr_upload_bytes(ent[2].data_ptr(), ent[0].data_ptr(), a.nbytes, self._stream()), 'spr_upload_bytes')
        return ent[2]

    _PINNED_RESULT_BYTES = 8 << 30      # page-locked memory handed out as results and still alive, at most (SPR_PINNED_RESULT_GB)

    def _pinned_result(self, shape, dtype):
        """A page-locked host tensor for a big result, or None when it cannot be had (then the pageable copy runs).
        PyTorch caches page-locked blocks: the first result of a size pays the pinning (59 ms for 720 MB), later ones
        reuse the blocks earlier results have returned.  Page-locked memory is a machine-wide resource: the results
        still referenced by the caller are counted (weak references to the ndarrays handed out, _export_pinned) and a new
        one is only pinned while the total stays under the budget."""
        torch = self.torch
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        budget = self.__dict__.get('_pinned_budget')
        if budget is None:                                     # SPR_PINNED_RESULT_GB, read once
            import os
            budget = self._PINNED_RESULT_BYTES
            env = os.environ.get('SPR_PINNED_RESULT_GB')
            if env is not None:
                try:
                    budget = int(float(env) * (1 << 30))
                except ValueError:
                    pass
            self._pinned_budget = budget
        live = getattr(self, '_pinned_live', None)
        if live is None:
            live = self._pinned_live = []
        live[:] = [(w, b) for w, b in live if w() is not None]
        if nbytes + sum(b for _, b in live) > budget:
            if budget > 0 and not self.__dict__.get('_pinned_warned'):
                # said once: from here on results come back as ordinary (pageable) arrays, at a fifth of the rate -- the caller is
                # holding on to more page-locked results than SPR_PINNED_RESULT_GB allows (pinning host RAM is a
                # machine-wide resource a library must not take without bound, nor give up silently)
                import warnings
                self._pinned_warned = True
                warnings.warn(f'openmeasure_amd: {sum(b for _, b in live) / 2 ** 30:.1f} GiB of page-locked results are still '
                              f'referenced; a further {nbytes / 2 ** 30:.1f} GiB would exceed SPR_PINNED_RESULT_GB = '
                              f'{budget / 2 ** 30:.0f}: this and later results are pageable copies (slower) until earlier ones are '
                              'released', RuntimeWarning, stacklevel=4)
            return None
        try:
            return torch.empty(shape, dtype=dtype, pin_memory=True)
        except RuntimeError:
            return None

    def _export_pinned(self, h):
        """The ndarray view of a page-locked result tensor, registered with the budget of _pinned_result: the array (and
        every view the caller takes of it) keeps the memory alive, so its own life is what is counted."""
        import weakref
        arr = h.numpy()
        self._pinned_live.append((weakref.ref(arr), h.numel() * h.element_size()))
        return arr

    def _to_host_big(self, t):
        """Results above _PINNED_MIN_BYTES (4 MiB) -- the (n, n_p) field reconstruct() returns to the caller, sparse_sensing.py:371-375 -- land in
        a page-locked tensor of their own, one asynchronous copy at the PCIe rate (57 GB/s against 10 GB/s into pageable
        memory, tools/transfer_probe.py); the ndarray handed back is that memory."""
        torch = self.torch
        h = self._pinned_result(tuple(t.shape), t.dtype)
        if h is None:
            return self._to_host_staged(t)
        h.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        ev.synchronize()
        return self._export_pinned(h)

    _STAGED_CHUNK = 64 << 20

    def _to_host_staged(self, t, out=None):
        """A result that may not be page-locked as a whole (the 46 GB basis of config 3 read as `spr.Ur`, a field beyond the
        budget): into an ordinary ndarray through two page-locked 64 MiB buffers -- DMA into one on a side stream while
        four threads copy the other out (28 GB/s against the 10 GB/s of a pageable tensor.cpu(), tools/transfer_probe.py).
        ``out``: a C-contiguous ndarray of t's shape and dtype to fill (e.g. a block of a larger result)."""
        torch = self.torch
        from concurrent.futures import ThreadPoolExecutor
        res = np.empty(tuple(t.shape), dtype=torch.empty((), dtype=t.dtype).numpy().dtype) if out is None else out
        if not t.is_contiguous() and t.dim() == 2 and t.shape[0] > 1:
            # a matrix with a padded row stride (the basis of an odd r is buf[:, :r]): row blocks, each made contiguous on its
            # own -- t.contiguous() of the whole would put a second copy of a 46 GB basis next to X
            rows = max(1, self._STAGED_CHUNK // max(1, t.shape[1] * t.element_size()))
            for i0 in range(0, t.shape[0], rows):
                self._to_host_staged(t[i0:i0 + rows].contiguous(), out=res[i0:i0 + rows])
            return res
        t = t.contiguous()
        flat = t.view(-1)
        n = flat.numel()
        dst = res.reshape(-1)
        if getattr(self, '_dstage2', None) is None:
            self._dstage2 = [torch.empty(self._STAGED_CHUNK, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self._copy_pool = ThreadPoolExecutor(4, thread_name_prefix='spr-d2h')
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(self.device)
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        self._side.wait_event(ready)
        ch = self._STAGED_CHUNK // flat.element_size()
        pend = [[], []]
        k = 0
        for i0 in range(0, n, ch):
            i1 = min(n, i0 + ch)
            for f in pend[k & 1]:                             # the slot's previous contents have been copied out
                f.result()
            stage = self._dstage2[k & 1][:(i1 - i0) * flat.element_size()].view(t.dtype)
            with torch.cuda.stream(self._side):
                stage.copy_(flat[i0:i1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._side)
            ev.synchronize()
            src = stage.numpy()
            q = -(-(i1 - i0) // 4)
            pend[k & 1] = [self._copy_pool.submit(np.copyto, dst[i0 + j * q:min(i1, i0 + (j + 1) * q)],
                                                  src[j * q:min(i1 - i0, (j + 1) * q)]) for j in range(4) if j * q < i1 - i0]
            k += 1
        for fs in pend:
            for f in fs:
                f.result()
        t.record_stream(self._side)
        return res

    def reconstruct_to_host(self, Ur, row0, n_points, n_features, rowmean, scale, A, chunks=8):
        """reconstruct() with the reference's output contract: the field as a HOST array.  -> ndarray (n_p, n) in
        page-locked memory, or None when that memory cannot be had / the result is small (callers then take
        reconstruct() + to_host()).  The rows go in `chunks` launches; each chunk's D2H copy runs on a side stream under
        the next chunk's kernel, so the call costs about max(kernel, PCIe) instead of their sum."""
        torch = self.torch
        n, r, ldu = self._check_matrix(Ur)
        n_p = A.shape[0]
        if n * n_p * 8 <= 4 * self._PINNED_MIN_BYTES or rowmean.shape[0] != n:
            return None
        host = self._pinned_result((n_p, n), torch.float64)
        if host is None:
            return None
        out = self.empty((n_p, n))
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(self.device)
        main = torch.cuda.current_stream(self.device)
        rows = -(-n // max(1, int(chunks)))
        rows = max(1 << 16, (rows + 4095) // 4096 * 4096)
        for i0 in range(0, n, rows):
            i1 = min(n, i0 + rows)
            self.reconstruct(Ur[i0:i1], row0 + i0, n_points, n_features, rowmean[i0:i1], scale, A, out=out[:, i0:i1])
            ev = torch.cuda.Event()
            ev.record(main)
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                for v in range(n_p):                          # contiguous pieces: a strided 2-D copy runs at a tenth of the rate
                    host[v, i0:i1].copy_(out[v, i0:i1], non_blocking=True)
        out.record_stream(self._side)
        self._side.synchronize()
        return self._export_pinned(host)

    def field_unstage(self, stage, out=None):
        """stage (world, n_p, n_loc) as the all-gather left it -> (n_p, world * n_loc), the vectors side by side
        (spr_field_unstage_f64)."""
        world, n_p, n_loc = stage.shape
        if out is None:
            out = self.empty((n_p, world * n_loc))
        _lib.check(self.lib.spr_field_unstage_f64(_ptr(stage.contiguous()), world, n_p, n_loc, _ptr(out), out.stride(0),
                                                  self._stream()), 'spr_field_unstage_f64')
        return out

    def field_unstage_blocks(self, stage, table, total):
        """stage (world, n_p, n_max) of blocks padded to n_max rows, table (world, 2) int64 = (offset in the result, rows) per
        rank -> (n_p, total), the blocks side by side (spr_field_unstage_blocks_f64)."""
        world, n_p, n_max = stage.shape
        out = self.empty((n_p, int(total)))
        lay = self.to_device(np.ascontiguousarray(table, dtype=np.int64).reshape(-1), dtype=self.torch.int64)
        _lib.check(self.lib.spr_field_unstage_blocks_f64(_ptr(stage.contiguous()), world, n_p, n_max, _ptr(lay), _ptr(out),
                                                         out.stride(0), self._stream()), 'spr_field_unstage_blocks_f64')
        return out

    def p2p_field_gather(self, world, rank, all_gather):
        """The CU-free field exchange of a sharded reconstruct() (openmeasure_amd/p2p.py); buffers are made by ensure()."""
        import os
        from .p2p import P2PFieldGather
        return P2PFieldGather(self, world, rank, all_gather, double_buffer=os.environ.get('SPR_P2P_BUFFERS') == '2',
                              loopback=int(os.environ.get('SPR_P2P_LOOPBACK', '0') or 0))

    # ---- collectives behind the C ABI (csrc/comm.hip): the library's own communicator over the RCCL already in the process ----
    def comm_create(self, world, rank, carry):
        """A communicator of libspr_hip.so for this rank on the current device (COLLECTIVE).  ``carry(bytes or None) -> bytes``:
        the caller's channel for rank 0's unique id (torch.distributed, MPI, a file) -- called with the id on rank 0, with
        None on the others; returns the id everywhere.  -> opaque handle for comm_allreduce / comm_allgather / fit_gram_pass."""
        import ctypes as C
        nb = int(self.lib.spr_comm_unique_id_bytes())
        buf = (C.c_ubyte * nb)()
        mine = None
        if rank == 0:
            _lib.check(self.lib.spr_comm_unique_id(buf), 'spr_comm_unique_id')
            mine = bytes(buf)
        got = carry(mine)
        if len(got) != nb:
            raise ValueError(f'comm_create: the unique id has {nb} bytes, the channel delivered {len(got)}')
        idb = (C.c_ubyte * nb).from_buffer_copy(got)
        comm = C.c_void_p()
        with self.torch.cuda.device(self.device):
            _lib.check(self.lib.spr_comm_init(idb, int(rank), int(world), C.byref(comm)), 'spr_comm_init')
        return comm

    def comm_destroy(self, comm):
        if comm is not None:
            _lib.check(self.lib.spr_comm_destroy(comm), 'spr_comm_destroy')

    def comm_allreduce(self, comm, t):
        """in-place sum over the ranks of a contiguous float64 / int64 tensor, on the current stream"""
        if not t.is_contiguous():
            raise ValueError('comm_allreduce: contiguous tensor')
        fn = {'torch.float64': self.lib.spr_allreduce_f64, 'torch.int64': self.lib.spr_allreduce_i64}.get(str(t.dtype))
        if fn is None:
            raise TypeError(f'comm_allreduce: float64 or int64, not {t.dtype}')
        if t.numel():
            _lib.check(fn(comm, _ptr(t), t.numel(), self._stream()), 'spr_allreduce')
        return t

    def comm_allgather(self, comm, t, out):
        """rank q's tensor t at out[q] on every rank (out: contiguous, world x t's bytes), on the current stream"""
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if nbytes:
            _lib.check(self.lib.spr_allgather(comm, _ptr(t), _ptr(out), nbytes, self._stream()), 'spr_allgather')
        return out

    def fit_gram_pass(self, X, row0, n_points, n_features, scale_type, comm, world):
        """The first pass of fit() with its collective as ONE library call (spr_fit_gram_pass): Gram kernel, finalize into the
        rank's slots of the collective buffer, all-reduce over ``comm`` (None: one rank), statistics merge + scaled sum.
        -> rowmean (n,), buf [F m m | world F 3 | world], packed [m m | 5 F], scale (F,), inv_scale (F,)"""
        n, m, ld = self._check_matrix(X)
        F = int(n_features)
        nbuf = int(self.lib.spr_fit_gram_pass_buffer(m, F, int(world))) // 8
        rowmean, buf = self.empty((n,)), self.empty((nbuf,))
        packed, scale, inv_scale = self.empty((m * m + 5 * F,)), self.empty((F,)), self.empty((F,))
        ws = self._workspace('gram', self.lib.spr_fit_gram_pass_workspace(m, F, n))
        tic, toc = self._timed('stats_gram')
        tic()
        _lib.check(self.lib.spr_fit_gram_pass(comm, _ptr(X), int(X.dtype == self.torch.float32), n, m, ld, row0, n_points, F,
                                              self.SCALE_CODES[scale_type], _ptr(rowmean), _ptr(buf), nbuf * 8, _ptr(packed),
                                              packed.data_ptr() + m * m * 8, _ptr(scale), _ptr(inv_scale), _ptr(ws), ws.numel(),
                                              self._stream()), 'spr_fit_gram_pass')
        toc()
        return rowmean, buf, packed, scale, inv_scale

    def stage_to_host(self, stage):
        """The same re-arrangement on the way to the host: block (q, v) of the staged field is copied straight to its place
        in a page-locked (n_p, world * n_loc) result -- no pass over the field on the device.  None: no pinned memory."""
        torch = self.torch
        world, n_p, n_loc = stage.shape
        host = self._pinned_result((n_p, world * n_loc), stage.dtype)
        if host is None:
            return None
        for q in range(world):
            for v in range(n_p):                              # contiguous pieces (see reconstruct_to_host)
                host[v, q * n_loc:(q + 1) * n_loc].copy_(stage[q, v], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        ev.synchronize()
        return self._export_pinned(host)

    def timing_event(self):
        """An event recorded NOW on the current stream; pairs are read with elapsed_ms() after a synchronisation."""
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(self.torch.cuda.current_stream(self.device))
        return ev

    @staticmethod
    def elapsed_ms(e0, e1):
        return e0.elapsed_time(e1)

    def _workspace(self, key, nbytes):
        cur = self._ws.get(key)
        if cur is None or cur.numel() < nbytes:
            cur = self.torch.empty(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)
            self._ws[key] = cur
        return cur

    def _check_matrix(self, X):
        t = self.torch
        if not (isinstance(X, t.Tensor) and X.is_cuda and X.dtype in (t.float64, t.float32) and X.dim() == 2
                and X.stride(1) == 1):
            raise TypeError('device matrix must be a 2-D float64 / float32 CUDA tensor with unit column stride')
        return X.shape[0], X.shape[1], X.stride(0)

    def _x(self, base, X):
        """entry point reading the snapshot shard X: <base>_f64, or <base>_x32 for an f32-stored shard"""
        return getattr(self.lib, base + ('_f64' if X.dtype == self.torch.float64 else '_x32'))

    def _u(self, base, Ur):
        """entry point reading the basis Ur: <base>_f64, or <base>_u32 for an f32-stored basis"""
        return getattr(self.lib, base + ('_f64' if Ur.dtype == self.torch.float64 else '_u32'))

    def time_next(self, kernel):
        """Bracket the next launch of `kernel` ('stats_gram' | 'project' | 'reconstruct') with a pair of
        timing events on the launch stream; returns (start, stop) torch events to read after a sync."""
        t = self.torch
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        self._timing[kernel] = (e0, e1)
        return e0, e1

    def _timed(self, kernel):
        ev = self._timing.pop(kernel, None)
        if ev is None:
            return (lambda: None), (lambda: None)
        st = self.torch.cuda.current_stream(self.device)
        return (lambda: ev[0].record(st)), (lambda: ev[1].record(st))

    # ---- K1 + K3a ------------------------------------------------------------------------
    def stats_gram(self, X, row0, n_points, n_features, center=True, gram_out=None, fstats_out=None):
        """-> rowmean (n,), fstats (F,3) = (count, mean, M2) of the local row means,
        gram (F,m,m) = per-feature sum of centred outer products over the local rows.
        ``gram_out`` / ``fstats_out``: contiguous float64 tensors of those shapes to write into (fit() hands over views of
        its collective buffer, so nothing is copied between the finalize kernel and the all-reduce); m <= 256 only."""
        n, m, ld = self._check_matrix(X)
        if m > _lib.SPR_MAX_M:
            rowmean, fstats, gram = self._stats_gram_wide(X, row0, n_points, n_features, center)
            if gram_out is not None:
                gram_out.copy_(gram)
                fstats_out.copy_(fstats)
                return rowmean, fstats_out, gram_out
            return rowmean, fstats, gram
        rowmean = self.empty((n,))
        fstats = fstats_out if fstats_out is not None else self.empty((n_features, 3))
        gram = gram_out if gram_out is not None else self.empty((n_features, m, m))
        if gram_out is not None and not (gram.is_contiguous() and fstats.is_contiguous() and gram.dtype == self.torch.float64
                                         and tuple(gram.shape) == (n_features, m, m) and tuple(fstats.shape) == (n_features, 3)):
            raise ValueError('stats_gram(gram_out, fstats_out): contiguous float64 (F, m, m) and (F, 3) tensors')
        nbytes = self.lib.spr_stats_gram_workspace(m, n_features)
        ws = self._workspace('gram', nbytes)
        tic, toc = self._timed('stats_gram')
        tic()
        _lib.check(self._x('spr_stats_gram', X)(_ptr(X), n, m, ld, row0, n_points, n_features, int(bool(center)),
                                               _ptr(rowmean), _ptr(ws), ws.numel(), self._stream()),
                   'spr_stats_gram_f64')
        toc()
        _lib.check(self.lib.spr_stats_gram_finalize_f64(n, m, row0, n_points, n_features, _ptr(ws), ws.numel(),
                                                        _ptr(fstats), _ptr(gram), m, 0, self._stream()),
                   'spr_stats_gram_finalize_f64')
        return rowmean, fstats, gram

    def _stats_gram_wide(self, X, row0, n_points, n_features, center):
        """m > 256: the columns go in slices of 256.  Two slices (m <= 512): A^T A (symmetric kernel, which forms the means
        of its slice), A^T B (cross kernel) and B^T B shifted by those same per-row constants, then P G P and the true means
        (spr_gram_shift_finish_f64).  More slices: a row-statistics pass, then every diagonal block and every slice pair --
        see spr_hip.h."""
        n, m, ld = self._check_matrix(X)
        if m > _lib.SPR_MAX_M_WIDE:
            return self._stats_gram_slices(X, row0, n_points, n_features, center)
        F, mA, st = n_features, _lib.SPR_MAX_M, self._stream()
        mB = m - mA
        fstats = self.zeros((F, 3))
        gram = self.empty((F, m, m))
        tic, toc = self._timed('stats_gram')
        tic()
        esz = X.element_size()
        wsx = self._workspace('cross', self.lib.spr_gram_cross_workspace(m, F))
        if not center:
            rowmean = self.zeros((n,))
            _lib.check(self._x('spr_gram_cross', X)(_ptr(X), n, m, ld, row0, n_points, F, 0, _ptr(rowmean), _ptr(gram),
                                                   _ptr(wsx), wsx.numel(), st), 'spr_gram_cross_f64')
            scratch = self.empty((F, 3))
            for origin, width in ((0, mA), (mA, mB)):
                ws = self._workspace('gram', self.lib.spr_stats_gram_workspace(width, F))
                xp = X.data_ptr() + origin * esz
                _lib.check(self._x('spr_stats_gram', X)(xp, n, width, ld, row0, n_points, F, 0, _ptr(rowmean), _ptr(ws),
                                                       ws.numel(), st), 'spr_stats_gram_f64')
                _lib.check(self.lib.spr_stats_gram_finalize_f64(n, width, row0, n_points, F, _ptr(ws), ws.numel(),
                                                                _ptr(scratch), _ptr(gram), m, origin, st),
                           'spr_stats_gram_finalize_f64')
            toc()
            return rowmean, fstats, gram
        # Centred: every launch shifts the rows by the mean of their FIRST 256 columns -- formed for free by the symmetric
        # launch on that slice -- and P G P afterwards gives the Gram matrix of the row-centred data (spr_hip.h): the cross
        # block, whose two workgroup flavours used to sum all 512 columns of every row for the full-row mean, only subtracts.
        rowmean, rowsum_b, scratch = self.empty((n,)), self.empty((n,)), self.empty((F, 3))
        ws = self._workspace('gram', self.lib.spr_stats_gram_workspace(mA, F))
        _lib.check(self._x('spr_stats_gram', X)(_ptr(X), n, mA, ld, row0, n_points, F, 1, _ptr(rowmean), _ptr(ws),
                                               ws.numel(), st), 'spr_stats_gram_f64')
        _lib.check(self.lib.spr_stats_gram_finalize_f64(n, mA, row0, n_points, F, _ptr(ws), ws.numel(), _ptr(scratch),
                                                        _ptr(gram), m, 0, st), 'spr_stats_gram_finalize_f64')
        _lib.check(self._x('spr_gram_cross', X)(_ptr(X), n, m, ld, row0, n_points, F, 2, _ptr(rowmean), _ptr(gram),
                                               _ptr(wsx), wsx.numel(), st), 'spr_gram_cross_f64')
        ws = self._workspace('gram', self.lib.spr_stats_gram_workspace(mB, F))
        _lib.check(self._x('spr_stats_gram_shifted', X)(X.data_ptr() + mA * esz, n, mB, ld, row0, n_points, F,
                                                       _ptr(rowmean), _ptr(rowsum_b), _ptr(ws), ws.numel(), st),
                   'spr_stats_gram_shifted_f64')
        _lib.check(self.lib.spr_stats_gram_finalize_f64(n, mB, row0, n_points, F, _ptr(ws), ws.numel(), _ptr(scratch),
                                                        _ptr(gram), m, mA, st), 'spr_stats_gram_finalize_f64')
        _lib.check(self.lib.spr_gram_shift_finish_f64(_ptr(rowmean), _ptr(rowsum_b), n, mA, m, _ptr(gram), F, st),
                   'spr_gram_shift_finish_f64')
        ws = self._workspace('rowstats', self.lib.spr_rowstats_workspace(F))
        _lib.check(self.lib.spr_rowmean_stats_f64(_ptr(rowmean), n, row0, n_points, F, _ptr(fstats), _ptr(ws),
                                                  ws.numel(), st), 'spr_rowmean_stats_f64')
        toc()
        return rowmean, fstats, gram

    def _stats_gram_slices(self, X, row0, n_points, n_features, center):
        """m > 512: q = ceil(m / 256) column slices -- one row-statistics pass (means of the full rows), q symmetric
        launches (diagonal blocks) and q (q - 1) / 2 slice-pair launches, all with external means."""
        n, m, ld = self._check_matrix(X)
        F, W, st = n_features, _lib.SPR_MAX_M, self._stream()
        esz = X.element_size()
        gram = self.empty((F, m, m))
        tic, toc = self._timed('stats_gram')
        tic()
        if center:
            rowmean, fstats = self.empty((n,)), self.empty((F, 3))
            ws = self._workspace('rowstats', self.lib.spr_rowstats_workspace(F))
            _lib.check(self._x('spr_rowstats', X)(_ptr(X), n, m, ld, row0, n_points, F, _ptr(rowmean), _ptr(fstats),
                                                 _ptr(ws), ws.numel(), st), 'spr_rowstats_f64')
        else:
            rowmean, fstats = self.zeros((n,)), self.zeros((F, 3))
        mode = 2 if center else 0
        scratch = self.empty((F, 3))
        slices = [(o, min(W, m - o)) for o in range(0, m, W)]
        for origin, width in slices:
            ws = self._workspace('gram', self.lib.spr_stats_gram_workspace(width, F))
            _lib.check(self._x('spr_stats_gram', X)(X.data_ptr() + origin * esz, n, width, ld, row0, n_points, F, mode,
                                                   _ptr(rowmean), _ptr(ws), ws.numel(), st), 'spr_stats_gram_f64')
            _lib.check(self.lib.spr_stats_gram_finalize_f64(n, width, row0, n_points, F, _ptr(ws), ws.numel(),
                                                            _ptr(scratch), _ptr(gram), m, origin, st),
                       'spr_stats_gram_finalize_f64')
        wsx = self._workspace('cross', self.lib.spr_gram_cross_workspace(2 * W, F))
        for i, (oa, _) in enumerate(slices):
            for ob, wb in slices[i + 1:]:
                _lib.check(self._x('spr_gram_cross_pair', X)(_ptr(X), n, oa, ob, wb, m, ld, row0, n_points, F, mode,
                                                            _ptr(rowmean), _ptr(gram), _ptr(wsx), wsx.numel(), st),
                           'spr_gram_cross_pair_f64')
        if not center:
            # un-centred statistics are not used by any caller (decomposition(X0) takes the Gram matrix only)
            fstats.zero_()
        toc()
        return rowmean, fstats, gram

    def gram_filler(self, X, rows, row0, n_points, n_features):
        """The Gram pass once more over the first `rows` rows of X, results discarded: real work of the real kernel, queued
        into the host gap of fit() so that the chip does not drop its clock while the host eigen-solves (see
        ROM.gap_filler).  Uses the Gram workspace (its slabs have been consumed by the finalize call in front) and a
        scratch vector for the row means."""
        n, m, ld = self._check_matrix(X)
        rows = int(min(rows, n))
        if rows <= 0:
            return
        w = min(m, _lib.SPR_MAX_M)                            # a wide X: its first 256-column slice (row stride = full row)
        scratch = self._workspace('fillmean', rows * 8)
        ws = self._workspace('gram', self.lib.spr_stats_gram_workspace(w, n_features))
        _lib.check(self._x('spr_stats_gram', X)(_ptr(X), rows, w, ld, row0, n_points, n_features, 1, scratch.data_ptr(),
                                               _ptr(ws), ws.numel(), self._stream()), 'spr_stats_gram_f64')

    # ---- K3b: device-side spectrum (m <= 64) ------------------------------------------------------
    SCALE_CODES = {'std': 0, 'none': 1, 'pareto': 2, 'vast': 3, 'level': 4, 'variance': 5, 'poisson': 6, 'l2-norm': 7}

    spectrum_max_sweeps = 15                                 # SP_MAX_SWEEPS of csrc/spectrum.hip (info[0] when not converged)

    @property
    def spectrum_max_m(self):
        return int(self.lib.spr_spectrum_max_m())

    def spectrum(self, gram, fstats_all, scale_type, r):
        """gram (F,m,m) all-reduced, fstats_all (ranks,F,3). -> dict of device tensors (see spr_hip.h)."""
        F, m = gram.shape[0], gram.shape[1]
        out = dict(feat=self.empty((F, 5)), scale=self.empty((F,)), inv_scale=self.empty((F,)), lam=self.empty((m,)),
                   S=self.empty((m,)), expvar=self.empty((m,)), V=self.empty((m, m)), W=self.empty((m, r)),
                   Ar=self.empty((m, r)), info=self.empty((3,)))
        _lib.check(self.lib.spr_spectrum_f64(_ptr(gram.contiguous()), _ptr(fstats_all.contiguous()),
                                             fstats_all.shape[0], F, m, self.SCALE_CODES[scale_type], r,
                                             _ptr(out['feat']), _ptr(out['scale']), _ptr(out['inv_scale']),
                                             _ptr(out['lam']), _ptr(out['S']), _ptr(out['expvar']), _ptr(out['V']),
                                             _ptr(out['W']), _ptr(out['Ar']), _ptr(out['info']), self._stream()),
                   'spr_spectrum_f64')
        return out

    def gram_combine(self, gram, fstats_all, scale_type):
        """gram (F,m,m) all-reduced, fstats_all (ranks,F,3) -> packed (m*m + 5F,) tensor [G | feat], scale (F,),
        inv_scale (F,) -- the statistics merge, feature scales and scaled sum of fit() on the device."""
        F, m = gram.shape[0], gram.shape[1]
        packed = self.empty((m * m + 5 * F,))
        scale, inv_scale = self.empty((F,)), self.empty((F,))
        _lib.check(self.lib.spr_gram_combine_f64(_ptr(gram.contiguous()), _ptr(fstats_all.contiguous()),
                                                 fstats_all.shape[0], F, m, self.SCALE_CODES[scale_type], _ptr(packed),
                                                 packed.data_ptr() + m * m * 8, _ptr(scale), _ptr(inv_scale),
                                                 self._stream()), 'spr_gram_combine_f64')
        return packed, scale, inv_scale

    # ---- K4 --------------------------------------------------------------------------------
    def _project_group(self, X, i0, rows, row0, n_points, n_features, inv_scale, Wg, rowmean, center, out_ptr, ldu,
                       out_f64, precenter, norms=None, force_stream=False):
        """One column group (q <= SPR_MAX_R columns of W, packed m x q) of the projection of rows [i0, i0+rows) of X,
        written at out_ptr with row stride ldu.  m <= 256: W-stationary / register-resident kernels (spr_project_*);
        wider X, or row means that must be removed before the multiplication: the streamed-W kernel, one launch over
        the full contraction length (spr_project_stream_*).  ``norms``: a (rows,) float64 tensor that receives the squared
        norms of the stored rows (spr_project_norms_* where the W-stationary kernel takes the shape, the streamed-W
        kernel for every other one)."""
        n, m, ld = self._check_matrix(X)
        q = Wg.shape[1]
        f32 = X.dtype == self.torch.float32
        xp = X.data_ptr() + i0 * ld * X.element_size()
        mean_p = rowmean.data_ptr() + i0 * rowmean.element_size() if center else None
        st = self._stream()
        sfx = '_f64' if not f32 else ('_x32_f64out' if out_f64 else '_x32')
        stream = m > _lib.SPR_MAX_M or (precenter and center) or self._force_stream or force_stream
        if norms is not None and not stream:
            stream = not self.lib.spr_project_norms_supported(m, q, rows, ld, xp, int(f32))
        if stream:
            name = ('spr_project_stream' if norms is None else 'spr_project_stream_norms') + sfx
            nbytes = self.lib.spr_project_stream_workspace(m, q, int(f32))
            ws = self._workspace('pstream', nbytes)
            extra = () if norms is None else (_ptr(norms),)
            _lib.check(getattr(self.lib, name)(xp, rows, m, ld, row0 + i0, n_points, n_features,
                                               (2 if precenter else 1) if center else 0, _ptr(inv_scale), mean_p,
                                               _ptr(Wg), q, out_ptr, ldu, *extra, _ptr(ws), ws.numel(), st), name)
        elif norms is not None:
            name = 'spr_project_norms' + sfx
            _lib.check(getattr(self.lib, name)(xp, rows, m, ld, row0 + i0, n_points, n_features, int(bool(center)),
                                               _ptr(inv_scale), mean_p, _ptr(Wg), q, out_ptr, ldu, _ptr(norms), st), name)
        else:
            name = 'spr_project' + sfx
            _lib.check(getattr(self.lib, name)(xp, rows, m, ld, row0 + i0, n_points, n_features, int(bool(center)),
                                               _ptr(inv_scale), mean_p, _ptr(Wg), q, out_ptr, ldu, 0, st), name)

    def project_writes_norms(self, X, r, center=True, precenter=False):
        """Does the projection kernel this shape takes ANYWAY produce row norms (project(norms=...) then costs no change
        of kernel)?  True for the W-stationary and the streamed-W kernel, False where the general kernel runs."""
        n, m, ld = self._check_matrix(X)
        if r > _lib.SPR_MAX_R:
            return False
        if m > _lib.SPR_MAX_M or (precenter and center) or self._force_stream:
            return True
        return bool(self.lib.spr_project_norms_supported(m, r, n, ld, X.data_ptr(), int(X.dtype == self.torch.float32)))

    def project(self, X, row0, n_points, n_features, inv_scale, W, center=True, out=None, rowmean=None,
                basis_dtype=None, precenter=False, norms=None):
        """Ur = ((X - rowmean) W) / X_scl ; W is (m,r) on the device. -> (n, r) tensor, row stride even.
        ``out``: a previous result of the same shape to overwrite (keeps one basis buffer alive).
        ``rowmean``: the row means from stats_gram (required when center=True).
        ``basis_dtype``: storage type of the result; default float64 (the reference's U is float64 whatever the dtype
        of X, sparse_sensing.py:106-107, :169, :272); torch.float32 only for a float32 X (storage option).
        ``precenter``: subtract the row means before the multiplication instead of in the epilogue.
        ``norms``: a (n,) float64 tensor to receive the squared norms of the rows as stored (r <= 128 only: one column
        group) -- what qr_begin(norms=...) starts a placement from without reading the basis again.
        Any m; r > 128 goes in column groups of 128 (each one more read of X)."""
        if center and rowmean is None:
            raise ValueError('project(center=True) needs the row means of the Gram pass')
        n, m, ld = self._check_matrix(X)
        r = W.shape[1]
        ldu = r + (r & 1)
        t = self.torch
        dt = basis_dtype or t.float64
        if dt == t.float32 and X.dtype != t.float32:
            raise ValueError('a float32 basis is a storage option of a float32 snapshot matrix')
        if (out is not None and tuple(out.shape) == (n, r) and out.stride(0) == ldu and out.stride(1) == 1
                and out.dtype == dt):
            buf = out
        else:
            del out
            buf = self.empty((n, ldu), dtype=dt)
        tic, toc = self._timed('project')
        tic()
        Wc = W.contiguous()
        G = _lib.SPR_MAX_R
        if norms is not None and (r > G or tuple(norms.shape) != (n,) or norms.dtype != t.float64):
            raise ValueError('project(norms=...): a float64 vector with one entry per row, r <= 128')
        # a float64 basis wider than 128 columns: groups of 256 through the streamed-W kernel's 16-tile form where the rows are
        # aligned (half the reads of X); everything else in groups of 128
        esz = X.element_size()
        if r > G and dt == t.float64 and m % 4 == 0 and (ld * esz) % 16 == 0 and X.data_ptr() % 16 == 0:
            G = _lib.SPR_MAX_R_STREAM
        for g0 in range(0, r, G):
            qg = min(G, r - g0)
            Wg = Wc if qg == r else Wc[:, g0:g0 + qg].contiguous()
            self._project_group(X, 0, n, row0, n_points, n_features, inv_scale, Wg, rowmean, center,
                                buf.data_ptr() + g0 * buf.element_size(), ldu, dt == t.float64, precenter, norms,
                                force_stream=qg > _lib.SPR_MAX_R)
        toc()
        return buf[:, :r] if buf.shape[1] != r else buf

    def project_f64(self, X, i0, rows, row0, n_points, n_features, inv_scale, W, rowmean, out, center=True,
                    precenter=False):
        """out[:rows, :q] = ((X[i0:i0+rows] - rowmean) W) / X_scl in float64 whatever the storage of X, for any
        q = W.shape[1] <= out.shape[1] (column groups of <= 128 go to column offsets of `out`).  Used by the
        conditioning refinement of fit(), whose product must not be rounded to a narrower storage type."""
        n, m, ld = self._check_matrix(X)
        q = W.shape[1]
        if not (out.dtype == self.torch.float64 and out.dim() == 2 and out.stride(1) == 1 and out.shape[1] >= q
                and out.shape[0] >= rows and out.stride(0) % 2 == 0):
            raise ValueError('project_f64: out must be a float64 matrix with an even row stride')
        Wc = W.contiguous()
        # up to 256 columns per launch in the streamed-W kernel's 16-tile form (float64 output, aligned rows, m % 4 == 0):
        # the refinement pass of fit() at m <= 256 then reads X ONCE per pass instead of once per 128-column group
        esz = X.element_size()
        wide = (m % 4 == 0 and (ld * esz) % 16 == 0 and (X.data_ptr() + i0 * ld * esz) % 16 == 0 and q > _lib.SPR_MAX_R)
        G = _lib.SPR_MAX_R_STREAM if wide else _lib.SPR_MAX_R
        for g0 in range(0, q, G):
            qg = min(G, q - g0)
            Wg = Wc if qg == q else Wc[:, g0:g0 + qg].contiguous()
            self._project_group(X, i0, rows, row0, n_points, n_features, inv_scale, Wg, rowmean, center,
                                out.data_ptr() + g0 * 8, out.stride(0), True, precenter, force_stream=qg > _lib.SPR_MAX_R)
        return out

    def feature_minmax(self, X, row0, n_points, n_features):
        """-> (F, 2) tensor: min and max of the raw block per feature over the local rows."""
        n, m, ld = self._check_matrix(X)
        out = self.empty((n_features, 2))
        ws = self._workspace('minmax', self.lib.spr_feature_minmax_workspace(n_features))
        _lib.check(self._x('spr_feature_minmax', X)(_ptr(X), n, m, ld, row0, n_points, n_features, _ptr(out), _ptr(ws),
                                                   ws.numel(), self._stream()), 'spr_feature_minmax_f64')
        return out

    def feature_digit_hist(self, X, row0, n_points, n_features, prefix, shift, bits, two_targets):
        """One radix-selection pass (spr_feature_digit_hist_f64).  prefix: (F, 2) int64 tensor holding the uint64 key
        prefixes; -> (F, 2, 1 << bits) int64 counts over the local rows."""
        n, m, ld = self._check_matrix(X)
        hist = self.zeros((n_features, 2, 1 << bits), dtype=self.torch.int64)
        _lib.check(self._x('spr_feature_digit_hist', X)(_ptr(X), n, m, ld, row0, n_points, n_features, _ptr(prefix),
                                                       shift, bits, int(bool(two_targets)), _ptr(hist),
                                                       self._stream()), 'spr_feature_digit_hist_f64')
        return hist

    def colsums(self, X, row0, n_points, n_features, rowmean):
        """-> (F, 2, m): sum_i c_i and sum_i mean_i c_i per feature (c_i = x_i - mean_i), local rows."""
        n, m, ld = self._check_matrix(X)
        out = self.empty((n_features, 2, m))
        ws = self._workspace('colsums', self.lib.spr_colsums_workspace(m, n_features))
        _lib.check(self._x('spr_colsums', X)(_ptr(X), n, m, ld, row0, n_points, n_features, _ptr(rowmean), _ptr(out),
                                            _ptr(ws), ws.numel(), self._stream()), 'spr_colsums_f64')
        return out

    def fill_feature(self, n_rows, row0, n_points, values):
        """-> (n_rows,) vector holding values[feature of the row]."""
        out = self.empty((n_rows,))
        _lib.check(self.lib.spr_fill_feature_f64(_ptr(out), n_rows, row0, n_points, values.shape[0], _ptr(values),
                                                 self._stream()), 'spr_fill_feature_f64')
        return out

    # ---- K2 / K11 stand-alone ---------------------------------------------------------------
    def scale_rows(self, X, row0, n_points, n_features, rowmean, inv_scale):
        n, m, ld = self._check_matrix(X)
        out = self.empty((n, m))
        _lib.check(self._x('spr_scale_rows', X)(_ptr(X), n, m, ld, row0, n_points, n_features, _ptr(rowmean),
                                               _ptr(inv_scale), _ptr(out), m, self._stream()),
                   'spr_scale_rows_f64')
        return out

    def unscale(self, x0, row0, n_points, n_features, rowmean, scale, rowscale=None):
        n = x0.shape[0]
        out = self.empty((n,))
        _lib.check(self.lib.spr_unscale_f64(_ptr(x0.contiguous()), n, row0, n_points, n_features, _ptr(rowmean),
                                            _ptr(scale), _ptr(rowscale), _ptr(out), self._stream()),
                   'spr_unscale_f64')
        return out

    # ---- K10 + K11 ---------------------------------------------------------------------------
    def reconstruct(self, Ur, row0, n_points, n_features, rowmean, scale, A, out=None, rowscale=None):
        """x = X_scl (Ur a) + X_cnt for the n_p rows of A. -> (n_p, n) tensor (column-major (n, n_p))."""
        n, r, ldu = self._check_matrix(Ur)
        n_p = A.shape[0]
        if out is None:
            out = self.empty((n_p, n))
        tic, toc = self._timed('reconstruct')
        tic()
        _lib.check(self._u('spr_reconstruct', Ur)(_ptr(Ur), n, r, ldu, row0, n_points, n_features, _ptr(rowmean),
                                                _ptr(scale), _ptr(rowscale), _ptr(A.contiguous()), n_p, _ptr(out),
                                                out.stride(0),
                                                self._stream()), 'spr_reconstruct_f64')
        toc()
        return out

    # ---- K6 ----------------------------------------------------------------------------------
    def mask_rows(self, Ur, mask_u8):
        n, r, ldu = self._check_matrix(Ur)
        _lib.check(self._u('spr_mask_rows', Ur)(_ptr(Ur), n, r, ldu, _ptr(mask_u8), self._stream()),
                   'spr_mask_rows_f64')

    @property
    def qr_batch(self):
        return int(self.lib.spr_qr_batch())

    def qr_begin(self, Ur, row0, n_steps, norms=None):
        """Allocate the pivoting state; initial norms, candidate set, local record and tau.  ``norms``: the squared row
        norms project(norms=...) left when it stored this very Ur -- the start then reads them (8 bytes per row) instead
        of sweeping the basis; the vector itself is not modified."""
        n, r, ldu = self._check_matrix(Ur)
        if r > _lib.SPR_MAX_R_WIDE:
            raise NotImplementedError(f'placement: r={r} modes exceed the built range (1..{_lib.SPR_MAX_R_WIDE})')
        t = self.torch
        # what the host reads after every batch of steps -- the steps' certification flags, the local record, tau -- lies in ONE
        # buffer: one small download per batch, no concatenation kernel in front of it
        flat = self.zeros((n_steps + r + 3 + 1,))
        st = dict(Ur=Ur, n=n, r=r, ldu=ldu, row0=row0,
                  nrm=self.empty((n,)), rec=flat[n_steps:n_steps + r + 3], tau=flat[n_steps + r + 3:],
                  Q=self.zeros((n_steps, r)), piv=self.zeros((n_steps,), dtype=t.int64),
                  gap=self.zeros((n_steps,)), ok=flat[:n_steps], flat=flat,
                  ws=self._workspace('qr', self.lib.spr_qr_workspace_r(n, r)))
        if norms is not None:
            if tuple(norms.shape) != (n,) or norms.dtype != t.float64:
                raise ValueError('qr_begin(norms=...): a float64 vector with one entry per row of Ur')
            _lib.check(self._u('spr_qr_init_norms', Ur)(_ptr(Ur), n, r, ldu, row0, _ptr(norms), _ptr(st['nrm']),
                                                      _ptr(st['rec']), _ptr(st['tau']), _ptr(st['ws']),
                                                      st['ws'].numel(), self._stream()), 'spr_qr_init_norms_f64')
            return st
        _lib.check(self._u('spr_qr_init', Ur)(_ptr(Ur), n, r, ldu, row0, _ptr(st['nrm']), _ptr(st['rec']),
                                            _ptr(st['tau']), _ptr(st['ws']), st['ws'].numel(), self._stream()),
                   'spr_qr_init_f64')
        return st

    def qr_step(self, st, step, recs, taus, first, xyz=None, n_points=0, d_min=0.0):
        """recs: (n_rank, r+3) records, taus: (n_rank, 1); one candidate-set step (see spr_hip.h).
        xyz (n_points, D) + d_min: GEM's distance exclusion around the step's pick."""
        _lib.check(self.lib.spr_qr_step_f64(st['n'], st['r'], step, _ptr(recs), recs.shape[0], _ptr(taus),
                                            taus.numel(), int(bool(first)), _ptr(st['Q']), _ptr(st['piv']),
                                            _ptr(st['gap']), _ptr(st['ok']), _ptr(st['rec']), _ptr(xyz),
                                            xyz.shape[1] if xyz is not None else 0, n_points, float(d_min),
                                            _ptr(st['ws']), st['ws'].numel(), self._stream()), 'spr_qr_step_f64')

    def qr_steps(self, st, step0, n_steps, xyz=None, n_points=0, d_min=0.0, first_exact=True):
        """Single GPU: n_steps consecutive candidate-set steps in one library call (spr_qr_steps_f64).
        first_exact=False after a pool sweep: the call's first step is certified against tau like the others."""
        _lib.check(self.lib.spr_qr_steps_f64(st['n'], st['r'], step0, n_steps, int(bool(first_exact)), _ptr(st['tau']),
                                             _ptr(st['Q']), _ptr(st['piv']), _ptr(st['gap']), _ptr(st['ok']),
                                             _ptr(st['rec']), _ptr(xyz), xyz.shape[1] if xyz is not None else 0, n_points,
                                             float(d_min), _ptr(st['ws']), st['ws'].numel(), self._stream()),
                   'spr_qr_steps_f64')

    # ---- K6, epoch sweeps (spr_qr_epoch_sweep_*, spr_qr_pool_build) -------------------------------------------------
    def qr_epoch_ok(self, st):
        """Can this basis take epoch sweeps (r a multiple of 16 up to 128, 16-byte rows, < 2^31 local rows)?"""
        Ur = st['Ur']
        return bool(self.lib.spr_qr_epoch_supported(st['r'], st['ldu'], Ur.data_ptr(), int(Ur.dtype == self.torch.float32),
                                                    st['n']))

    def qr_epoch_max_directions(self, st):
        """Directions one epoch sweep can apply (the direction tiles its LDS image holds)."""
        return int(self.lib.spr_qr_epoch_max_directions(st['r']))

    def qr_epoch_begin(self, st):
        """Epoch 0 starts from the exact initial norms."""
        st['nrm_e'] = st['nrm'].clone()
        st['pool_n'] = 0

    def qr_pool_build(self, st, theta):
        """Pool of the epoch: the local rows with nrm_e > theta, ascending.  -> their number (one host sync), -1 when the
        list would not fit (more than a quarter of the rows: no pool then)."""
        t = self.torch
        cap = max(st['n'] // 4, 1 << 16)
        if 'pool' not in st:
            st['pool'] = self.empty((cap,), dtype=t.int32)
            st['pool_cnt'] = self.zeros((1,), dtype=t.int32)
        ws = self._workspace('qrpool', self.lib.spr_qr_pool_workspace())
        _lib.check(self.lib.spr_qr_pool_build(_ptr(st['nrm_e']), st['n'], float(theta), _ptr(st['pool']), cap,
                                              _ptr(st['pool_cnt']), _ptr(ws), ws.numel(), self._stream()),
                   'spr_qr_pool_build')
        st['pool_n'] = int(self.to_host(st['pool_cnt'])[0])
        return st['pool_n']

    def qr_epoch_sweep(self, st, j_e, j, j_mark, pool=False, tau_floor=-2.0):
        """Residual norms from the epoch norms and the directions Q[j_e:j]: for the pool's rows (pool=True; steps are then
        certified against max(tau, tau_floor)) or for every row (pool=False: also rewrites the epoch norms, the next epoch
        starts at j); candidates / record / tau redrawn either way.  piv[j_mark:j] leave the race."""
        pl = st['pool'] if pool else None
        _lib.check(self._u('spr_qr_epoch_sweep', st['Ur'])(_ptr(st['Ur']), st['n'], st['r'], st['ldu'], st['row0'],
                                                          _ptr(st['Q']), _ptr(st['piv']), j_e, j, j_mark, _ptr(st['nrm_e']),
                                                          _ptr(st['nrm']), _ptr(pl), _ptr(st['pool_cnt']) if pool else None,
                                                          st['pool_n'] if pool else 0, float(tau_floor), _ptr(st['rec']),
                                                          _ptr(st['tau']), _ptr(st['ws']), st['ws'].numel(),
                                                          self._stream()), 'spr_qr_epoch_sweep_f64')

    def qr_exclude(self, st, mask=None, xyz=None, n_points=1, j0=0, nq=0, d_min=0.0):
        """Rows outside `mask` (uint8 per local row) and rows closer than d_min to the picks piv[j0:j0+nq] leave
        the pool (spr_qr_exclude_f64)."""
        piv = st['piv'][j0:j0 + nq] if nq else None
        _lib.check(self.lib.spr_qr_exclude_f64(_ptr(st['nrm']), st['n'], st['row0'], n_points, _ptr(mask), _ptr(xyz),
                                               xyz.shape[1] if xyz is not None else 0, _ptr(piv), nq, float(d_min),
                                               self._stream()), 'spr_qr_exclude_f64')

    def qr_refresh(self, st, j0, nq):
        """Apply the accepted directions Q[j0:j0+nq] to every row, redraw candidates / record / tau."""
        _lib.check(self._u('spr_qr_refresh', st['Ur'])(_ptr(st['Ur']), st['n'], st['r'], st['ldu'], st['row0'],
                                               _ptr(st['Q']), _ptr(st['piv']), j0, nq, _ptr(st['nrm']),
                                               _ptr(st['rec']), _ptr(st['tau']), _ptr(st['ws']),
                                               st['ws'].numel(), self._stream()), 'spr_qr_refresh_f64')

    def qr_apply(self, st, dirs, picks):
        """Down-date every row with the given directions (k, r) -- nrm <- max(nrm - (u.d)^2, 0) per direction --, take
        the rows `picks` (k int64 global rows, -1 = none) out of the pool and redraw candidates / record / tau.  Same
        sweep kernel as qr_refresh, for directions that did not come out of the candidate steps (GEM's ridge phase)."""
        k = dirs.shape[0]
        t = self.torch
        for j0 in range(0, k, self.qr_batch):
            nq = min(self.qr_batch, k - j0)
            Qv = dirs[j0:j0 + nq].contiguous()
            pv = picks[j0:j0 + nq].contiguous().to(t.int64)
            _lib.check(self._u('spr_qr_refresh', st['Ur'])(_ptr(st['Ur']), st['n'], st['r'], st['ldu'], st['row0'],
                                                          _ptr(Qv), _ptr(pv), 0, nq, _ptr(st['nrm']), _ptr(st['rec']),
                                                          _ptr(st['tau']), _ptr(st['ws']), st['ws'].numel(),
                                                          self._stream()), 'spr_qr_refresh_f64')

    # ---- K7 + K8 -------------------------------------------------------------------------------
    def measure_csr(self, indptr, indices, vals, Ur, row0, rowmean, scale=None, n_points=0):
        """-> Theta (s,r), cnt (s,) [, scl (s,) = C . X_scl when the per-feature scale is given]."""
        n, r, ldu = self._check_matrix(Ur)
        s = indptr.shape[0] - 1
        Theta = self.empty((s, r))
        cnt = self.empty((s,))
        scl = self.empty((s,)) if scale is not None else None
        _lib.check(self._u('spr_measure_csr', Ur)(_ptr(indptr), _ptr(indices), _ptr(vals), s, _ptr(Ur), n, r, ldu,
                                                row0, _ptr(rowmean), _ptr(scale), n_points,
                                                scale.shape[0] if scale is not None else 0, _ptr(Theta), _ptr(cnt),
                                                _ptr(scl), self._stream()), 'spr_measure_csr_f64')
        return (Theta, cnt) if scale is None else (Theta, cnt, scl)

    # ---- K8 + K9 -------------------------------------------------------------------------------
    ols_max_r = _lib.SPR_MAX_R_WIDE

    def _solve_outputs(self, n_p, s, r, n_info):
        """info (n_p, n_info), Ar (n_p, r), Ar_sigma (n_p, r), y0 (n_p, s, 2) as views of ONE buffer: predict() reads all four on the
        host, and one download of the buffer (to_host_views) costs a quarter of four (predict at config 3: 0.33 -> 0.2 ms)."""
        sizes = (n_p * n_info, n_p * r, n_p * r, n_p * s * 2)
        pad = [-(-k // 32) * 32 for k in sizes]                # every piece starts on a 256-byte boundary, like an allocation of its own
        flat = self.empty((sum(pad),))
        o = np.cumsum([0] + pad)
        return (flat[o[0]:o[0] + sizes[0]].view(n_p, n_info), flat[o[1]:o[1] + sizes[1]].view(n_p, r),
                flat[o[2]:o[2] + sizes[2]].view(n_p, r), flat[o[3]:o[3] + sizes[3]].view(n_p, s, 2))

    def to_host_views(self, *tensors):
        """Host copies of several device tensors; ONE download when they are views of one buffer (see _solve_outputs)."""
        base = tensors[0]._base if tensors else None
        if base is None or any(t._base is not base or not t.is_contiguous() for t in tensors) or base.dim() != 1:
            return tuple(self.to_host(t) for t in tensors)
        host = self.to_host(base)
        return tuple(host[t.storage_offset() - base.storage_offset():t.storage_offset() - base.storage_offset() + t.numel()]
                     .reshape(tuple(t.shape)).copy() for t in tensors)

    def solve_ols(self, Theta, cnt, scale, y):
        """y: (n_p, s, 3) device tensor. -> Ar (n_p,r), Ar_sigma (n_p,r), y0 (n_p,s,2), info (n_p,2)."""
        s, r = Theta.shape
        n_p = y.shape[0]
        info, Ar, Ar_sigma, y0 = self._solve_outputs(n_p, s, r, 2)
        if r > _lib.SPR_MAX_R:                                # matrices in a workspace instead of LDS
            ws = self._workspace('ols', self.lib.spr_solve_ols_workspace(s, r, n_p))
            _lib.check(self.lib.spr_solve_ols_wide_f64(_ptr(Theta.contiguous()), s, r, _ptr(cnt), _ptr(scale),
                                                       scale.shape[0], _ptr(y.contiguous()), n_p, _ptr(Ar),
                                                       _ptr(Ar_sigma), _ptr(y0), _ptr(info), _ptr(ws), ws.numel(),
                                                       self._stream()), 'spr_solve_ols_wide_f64')
            return Ar, Ar_sigma, y0, info
        _lib.check(self.lib.spr_solve_ols_f64(_ptr(Theta.contiguous()), s, r, _ptr(cnt), _ptr(scale),
                                              scale.shape[0], _ptr(y.contiguous()), n_p, _ptr(Ar), _ptr(Ar_sigma),
                                              _ptr(y0), _ptr(info), self._stream()), 'spr_solve_ols_f64')
        return Ar, Ar_sigma, y0, info

    def solve_pinv(self, Theta, cnt, scale, y, rcond=1e-15):
        """Minimum-norm least squares with np.linalg.pinv's semantics (spr_solve_pinv_f64): same arguments as
        solve_ols. -> Ar (n_p,r), Ar_sigma (n_p,r), y0 (n_p,s,2), info (n_p,4) = sweeps, rank, sigma_max, sigma_min."""
        s, r = Theta.shape
        n_p = y.shape[0]
        info, Ar, Ar_sigma, y0 = self._solve_outputs(n_p, s, r, 4)
        if r > _lib.SPR_MAX_R:                                # factor in a workspace instead of LDS
            if r > _lib.SPR_MAX_R_WIDE:
                raise NotImplementedError(f'solve: r={r} modes exceed the built range (1..{_lib.SPR_MAX_R_WIDE})')
            ws = self._workspace('pinv', self.lib.spr_solve_pinv_workspace(r, n_p))
            _lib.check(self.lib.spr_solve_pinv_wide_f64(_ptr(Theta.contiguous()), s, r, _ptr(cnt), cnt.shape[0],
                                                        _ptr(scale), scale.shape[0], _ptr(y.contiguous()), n_p,
                                                        float(rcond), _ptr(Ar), _ptr(Ar_sigma), _ptr(y0), _ptr(info),
                                                        _ptr(ws), ws.numel(), self._stream()), 'spr_solve_pinv_wide_f64')
            return Ar, Ar_sigma, y0, info
        _lib.check(self.lib.spr_solve_pinv_f64(_ptr(Theta.contiguous()), s, r, _ptr(cnt), cnt.shape[0], _ptr(scale),
                                               scale.shape[0], _ptr(y.contiguous()), n_p, float(rcond), _ptr(Ar),
                                               _ptr(Ar_sigma), _ptr(y0), _ptr(info), self._stream()),
                   'spr_solve_pinv_f64')
        return Ar, Ar_sigma, y0, info

    # ---- synthetic data ---------------------------------------------------------------------------
    def synth(self, n_rows, m, row0, n_points, R, eps, seed, out=None, dtype=None):
        """Rows [row0, row0+n_rows) of the synthetic matrix; R is (k, >=m) on the device.  dtype float32 stores
        the same f64 values rounded once."""
        if out is None:
            out = self.empty((n_rows, m), dtype=dtype)
        k, ldr = R.shape[0], R.stride(0)
        fn = self.lib.spr_synth_f64 if out.dtype == self.torch.float64 else self.lib.spr_synth_f32
        _lib.check(fn(_ptr(out), n_rows, m, out.stride(0), row0, n_points, 0, _ptr(R), k, ldr,
                                          float(eps), int(seed), self._stream()), 'spr_synth_f64')
        return out

    def synth_gather(self, rows, n_points, col, R, eps, seed):
        out = self.empty((rows.shape[0],))
        _lib.check(self.lib.spr_synth_gather_f64(_ptr(rows), rows.shape[0], n_points, col, _ptr(R), R.shape[0],
                                                 R.stride(0), float(eps), int(seed), _ptr(out), self._stream()),
                   'spr_synth_gather_f64')
        return out
