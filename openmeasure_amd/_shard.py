"""Row sharding of ROM / SPR objects (BASELINE north_star: "row-sharded across the 8 GPUs of one node with a single RCCL
all-reduce for the Gram matrix and a final all-gather for the reconstructed field"; the reference has no counterpart): the
RowShard descriptor, PendingField, and the methods of ROM that issue collectives or run the field exchange (a mixin:
rom.ROM inherits them) -- block layout, all-reduce / all-gather / broadcast through torch.distributed or the library's own
communicator, agreement of the ranks' host factors, the p2p / RCCL field exchange with its first-exchange checks and trial."""
from __future__ import annotations

import numpy as np

class RowShard:
    """This process's slice of the global feature-major snapshot matrix.

    row0      first global row held locally
    n_global  total rows (= n_points * n_features) over all ranks
    group     torch.distributed process group (None = default group); world size 1 if
              torch.distributed is not initialised.
    The ranks' blocks are contiguous, in rank order, and together cover all n_global rows (checked at the first gather);
    equal blocks (n_global / world rows each) take the zero-copy field all-gather, unequal ones are padded to the
    largest block for the gather and packed afterwards.
    broadcast_basis  every rank eigen-solves the same all-reduced Gram matrix (RCCL leaves identical bits on all
              ranks) with the same single-threaded LAPACK, so on one node the factors agree bit for bit and nothing
              is exchanged; set True when the ranks' hosts may differ (CPU type, LAPACK build): rank 0's
              decomposition is then broadcast, at the price of one more round trip per fit.
    partial   the ranks of the group together hold only a slice of the global rows (one rank's block of a larger job
              run on its own, bench.py --share-of): the global numbering still places the feature boundaries, the
              statistics, the basis and the gathered field are those of the rows the group holds.
    gather    how reconstruct() brings every rank's block of the field to every rank: 'rccl' -- torch.distributed's
              all-gather (a device kernel); 'p2p' -- the ranks of ONE node map each other's copy of the field and push
              their block into it with the SDMA engines (openmeasure_amd/p2p.py: no compute unit, so a gather left in
              flight really runs under the next fit(); any block sizes without padding); 'auto' (default) -- 'p2p' when
              its collective self-test passes on every rank, else 'rccl', with the reason on stderr and in
              ``rom.gather_path_``.  SPR_GATHER=rccl|p2p|auto overrides.  With 'auto' the first full-size exchange also times
              both paths under a Gram pass and keeps the faster (ROM._gather_trial; SPR_GATHER_TRIAL=0: p2p whenever available).
    native_comm  the all-reduces and all-gathers run through libspr_hip.so's OWN communicator (include/spr_hip.h: spr_comm_*,
              spr_fit_gram_pass -- fit()'s Gram pass, all-reduce and statistics merge as one enqueue) over the RCCL library that
              is already in the process; torch.distributed then only carries the communicator's unique id (and ``group`` says
              who takes part).  Off by default (SPR_NATIVE_COMM=1 switches it on): the same bits either way, and the default
              path is the one the gloo tests cover; needs one GPU per rank like any RCCL communicator.
    """

    def __init__(self, row0, n_global, group=None, force_collectives=False, broadcast_basis=False, partial=False,
                 gather='auto', native_comm=False):
        if gather not in ('auto', 'p2p', 'rccl'):
            raise ValueError("gather must be 'auto', 'p2p' or 'rccl'")
        self.native_comm = bool(native_comm)
        self.row0 = int(row0)
        self.n_global = int(n_global)
        self.group = group
        self.force_collectives = bool(force_collectives)   # issue the collectives even in a 1-rank group (tests)
        self.broadcast_basis = bool(broadcast_basis)
        self.partial = bool(partial)
        self.gather = gather

    @property
    def world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self):
        import torch.distributed as dist
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0


class PendingField:
    """Result of ``reconstruct(..., to_host=False, wait=False)``: the (n_p, n) field in HBM whose exchange between the
    ranks may still be in flight (RCCL's all-gather on its communication stream, or the SDMA pushes of the p2p path) -- or,
    with ``ROM.defer_reconstruct`` (the default), whose kernel has not even been launched yet (``launch``: it runs in the host gap of the
    object's next fit(), or here).  ``wait()`` makes the current stream wait for it and returns the tensor; nothing else may
    read the tensor before that.
    ``needs_cus``: the exchange runs a device kernel (RCCL) and competes with the caller's kernels for compute units."""

    def __init__(self, tensor, works=(), keep=(), on_wait=None, join=None, needs_cus=True, launch=None, shape=None):
        self._tensor = tensor
        self._works = list(works)
        self._join = join                      # p2p path: enqueues the stream waits on the arrival counters
        self._keep = keep                      # the gather's source buffers stay alive until it has been joined
        self._on_wait = on_wait                # ROM.comm_timing: brackets the join with two stream events
        self._launch = launch                  # deferred: () -> tensor or PendingField
        self._inner = None
        self._shape = tuple(shape) if shape is not None else None
        self._needs_cus = bool(needs_cus)

    @property
    def needs_cus(self):
        if self._inner is not None:
            return self._inner.needs_cus
        return self._needs_cus and (bool(self._works) or self._join is not None)

    @property
    def shape(self):
        return self._shape if self._tensor is None else tuple(self._tensor.shape)

    @property
    def launched(self):
        return self._launch is None

    @property
    def pending(self):
        """True while the field has not been launched or its exchange has not been joined."""
        if self._launch is not None:
            return True
        if self._inner is not None:
            return self._inner.pending
        return bool(self._works) or self._join is not None

    def launch(self):
        """Enqueue a deferred reconstruct now (no-op otherwise)."""
        fn, self._launch = self._launch, None
        if fn is not None:
            res = fn()
            if isinstance(res, PendingField):
                self._inner = res
            else:
                self._tensor = res

    def wait(self):
        self.launch()
        if self._inner is not None:
            self._tensor = self._inner.wait()
            self._inner = None
            return self._tensor
        done = self._on_wait() if ((self._works or self._join is not None) and self._on_wait is not None) else None
        for w in self._works:
            w.wait()
        if self._join is not None:
            self._join()
        if done is not None:
            done()
        self._works = []
        self._join = None
        self._keep = ()
        self._on_wait = None
        return self._tensor


class ShardedOps:
    """What a ROM does across ranks (mixed into rom.ROM; every method here expects ROM's attributes)."""

    def _world(self):
        return self._shard.world if self._shard is not None else 1

    def _dist(self):
        """True when collectives have to be issued (more than one rank, or forced for testing)."""
        return self._shard is not None and (self._shard.world > 1 or self._shard.force_collectives)

    def _shard_layout(self, n_loc):
        """(first row, rows) of every rank's block, rank order, cached.  Blocks that do not tile the global rows raise
        ValueError -- unless the shard is a declared slice of a larger job (partial)."""
        lay = self.__dict__.get('_layout')
        if lay is None:
            eng = self._engine()
            src = self.__dict__.pop('_layout_src', None)
            if src is not None:
                # fit()'s ONE all-reduce already carried every rank's row count per feature (the statistics slots) and
                # its first row: no collective here, and every rank sees the same table (they raise together)
                rows = np.rint(eng.to_host(src[0].sum(1))).astype(np.int64)
                lay = np.stack([np.rint(eng.to_host(src[1])).astype(np.int64), rows], axis=1)
            else:                                             # a fit route without those slots: ask (two integers per rank)
                mine = eng.to_device(np.array([self._row0, n_loc], dtype=np.float64))  # exact below 2^53
                lay = eng.to_host(self._all_gather(mine)).astype(np.int64)
            self._layout = lay
            if not self._shard.partial:
                ends = lay[:, 0] + lay[:, 1]
                if lay[0, 0] != 0 or ends[-1] != self._n_global or np.any(lay[1:, 0] != ends[:-1]):
                    raise ValueError('The row blocks of the ranks are not contiguous in rank order or do not cover the '
                                     f'{self._n_global} global rows: (first row, rows) per rank = {lay.tolist()}')
        return lay

    #: (name, shape, time.time()) of the last collective this object ENTERED -- what a watchdog prints when a rank hangs
    last_comm_ = None

    def _native_comm(self):
        """libspr_hip.so's own communicator (RowShard(native_comm=True) / SPR_NATIVE_COMM=1), created at first use -- COLLECTIVE:
        rank 0's unique id travels through ONE torch.distributed broadcast -- or None: the collectives go through
        torch.distributed."""
        if not self._dist():
            return None
        nc = self.__dict__.get('_ncomm')
        if nc is not None:
            return nc or None
        import os
        eng = self._engine()
        env = os.environ.get('SPR_NATIVE_COMM')
        if not ((self._shard.native_comm or env == '1') and env != '0') or not hasattr(eng, 'comm_create'):
            self._ncomm = False
            return None
        import time
        import torch.distributed as dist
        t = eng.torch
        nb = int(eng.lib.spr_comm_unique_id_bytes())

        def carry(idb):
            buf = t.zeros(nb, dtype=t.uint8) if idb is None else t.tensor(list(idb), dtype=t.uint8)
            if dist.get_backend(self._shard.group) == 'nccl':
                buf = buf.to(eng.device)
            src = dist.get_global_rank(self._shard.group, 0) if self._shard.group is not None else 0
            self.last_comm_ = ('broadcast (unique id of the native communicator)', (nb,), time.time())
            dist.broadcast(buf, src=src, group=self._shard.group)
            return bytes(buf.cpu().numpy().tobytes())
        self._ncomm = eng.comm_create(self._world(), self._shard.rank, carry)
        self.comm_library_ = eng.lib.spr_comm_library().decode()
        return self._ncomm

    def _all_reduce(self, t):
        if self._dist():
            import time
            self.last_comm_ = ('all_reduce', tuple(t.shape), time.time())
            nc = self._native_comm()
            if nc is not None and t.is_contiguous() and str(t.dtype) in ('torch.float64', 'torch.int64'):
                return self._engine().comm_allreduce(nc, t)
            import torch.distributed as dist
            dist.all_reduce(t, group=self._shard.group)
        return t

    def _all_gather(self, t):
        """-> tensor (world, *t.shape)"""
        if not self._dist():
            return t[None]
        import time
        self.last_comm_ = ('all_gather', tuple(t.shape), time.time())
        flat = t.contiguous().view(-1)                       # concatenated layout: accepted by RCCL and gloo alike
        out = flat.new_empty((self._world() * flat.numel(),))
        nc = self._native_comm()
        if nc is not None:
            self._engine().comm_allgather(nc, flat, out)
        else:
            import torch.distributed as dist
            dist.all_gather_into_tensor(out, flat, group=self._shard.group)
        return out.view((self._world(),) + tuple(t.shape))

    def _broadcasts_basis(self):
        """Do the ranks take rank 0's host factors instead of their own?  RowShard(broadcast_basis=True), or the check of
        the first fit() found ranks whose eigen-solves differ (_factors_agree)."""
        return self._dist() and (self._shard.broadcast_basis or self.__dict__.get('_basis_diverged', False))

    def _bcast(self, pack):
        """rank 0's float64 vector `pack` (same length on every rank) -> every rank"""
        import torch.distributed as dist
        eng = self._engine()
        nc = self._native_comm()
        if nc is not None:                                    # a sum in which only rank 0 contributes (adding zeros is exact)
            t = eng.to_device(pack if self._shard.rank == 0 else np.zeros_like(pack))
            return eng.to_host(eng.comm_allreduce(nc, t))
        t = eng.to_device(pack)
        dist.broadcast(t, src=dist.get_global_rank(self._shard.group, 0) if self._shard.group is not None else 0,
                       group=self._shard.group)
        return eng.to_host(t)

    def _same_on_all_ranks(self, *arrays):
        """rank 0's host arrays win (shapes agree on all ranks); identity unless the basis is broadcast"""
        if not self._broadcasts_basis():
            return arrays
        pack = self._bcast(np.concatenate([np.ravel(a) for a in arrays]))
        out, o = [], 0
        for a in arrays:
            out.append(pack[o:o + a.size].reshape(a.shape).copy())
            o += a.size
        return tuple(out)

    def _factors_agree(self, lam, V):
        """First sharded fit(): every rank has eigen-solved the same all-reduced Gram matrix on its own host.  On one node
        with one LAPACK the results agree bit for bit and nothing needs to be exchanged -- but a node whose sockets or
        libraries differ would let the ranks project onto slightly different bases WITHOUT any error.  One all-gather of a
        64-bit digest of (lam, V) per rank settles it; every rank sees the same table, hence the same verdict."""
        import hashlib
        eng = self._engine()
        h = hashlib.blake2b(np.ascontiguousarray(lam).tobytes() + np.ascontiguousarray(V).tobytes(), digest_size=8).digest()
        mine = np.array([int.from_bytes(h[:4], 'little'), int.from_bytes(h[4:], 'little'), V.shape[1]], dtype=np.float64)
        table = eng.to_host(self._all_gather(eng.to_device(mine)))
        return bool(np.all(table == table[0]))

    def _broadcast_factors(self, G, rank_of, have):
        """Rank 0's (lam, V) on every rank; ``have``: what this rank has already computed (rank 0 re-uses it).  Rank 0's
        LinAlgError travels in the header, so all ranks raise together."""
        m = G.shape[0]
        pack = np.zeros(2 + m + m * m)
        if self._shard.rank == 0:
            try:
                lam, V = have if have is not None else self._eig_local(G, rank_of)
                pack[0], pack[1] = 1.0, V.shape[1]
                pack[2:2 + m] = lam
                pack[2 + m:2 + m + m * V.shape[1]] = V.ravel()
            except np.linalg.LinAlgError:
                pack[0] = -1.0
        pack = self._bcast(pack)
        if pack[0] < 0:
            raise np.linalg.LinAlgError('Eigenvalues did not converge')
        k = int(pack[1])
        return pack[2:2 + m].copy(), pack[2 + m:2 + m + m * k].reshape(m, k).copy()

    # ------------------------------------------------------------------ the field exchange of sharded objects
    def _gather_select(self, n_p, lay):
        """'p2p' or 'rccl' for this object's field exchange (RowShard.gather / SPR_GATHER), decided at the first sharded
        reconstruct() -- by all ranks together: the p2p set-up ends with a collective self-test whose verdict every rank
        shares, so no rank can take one path while its peers take the other; with 'auto' the first full-size exchange then
        times both paths under a Gram pass and keeps the faster (_gather_trial).  ``gather_path_`` says what was chosen and why."""
        sel = self.__dict__.get('_gather_sel')
        if sel is not None:
            return sel
        import os
        import sys
        eng = self._engine()
        want = os.environ.get('SPR_GATHER') or self._shard.gather
        if want not in ('auto', 'p2p', 'rccl'):
            raise ValueError(f"SPR_GATHER={want!r}: 'auto', 'p2p' or 'rccl'")
        if want == 'rccl':
            sel, why = 'rccl', 'rccl (asked for)'
        elif not hasattr(eng, 'p2p_field_gather'):
            if want == 'p2p':
                raise RuntimeError("RowShard(gather='p2p'): this engine has no p2p field exchange")
            sel, why = 'rccl', 'rccl (engine without p2p exchange)'
        else:
            from .p2p import P2PUnavailable
            px = self.__dict__.get('_p2p')
            try:
                if px is None:
                    px = eng.p2p_field_gather(self._world(), self._shard.rank, self._all_gather)
                px.ensure(n_p, int(lay[:, 1].sum()))
                self._p2p = px
                sel, why = 'p2p', 'p2p (SDMA pushes into peer-mapped buffers, no compute units)'
            except P2PUnavailable as exc:
                if want == 'p2p':
                    raise
                sel, why = 'rccl', f'rccl (p2p unavailable: {exc})'
                if self._shard.rank == 0:
                    print(f'[openmeasure_amd] field exchange falls back to the RCCL all-gather: {exc}', file=sys.stderr)
        self._gather_sel = sel
        self.gather_path_ = why
        return sel

    def close(self):
        """Give back what a sharded object holds outside PyTorch's allocator: the persistent copy of the field and the counter
        page of the p2p exchange, mapped by the peers (COLLECTIVE: every rank calls it; the ranks meet between unmapping and
        freeing).  The object can be used again afterwards (the buffers are set up anew).  Without it they live until the process
        ends -- an interprocess mapping cannot be torn down from a finaliser."""
        self._flush_deferred()
        pf = self.__dict__.get('_pending_field')
        if pf is not None and pf.pending:
            pf.wait()
        px = self.__dict__.pop('_p2p', None)
        if px is not None:
            px.close()
        self.__dict__.pop('_gather_sel', None)
        nc = self.__dict__.pop('_ncomm', None)
        if nc:
            self._engine().torch.cuda.synchronize(self._engine().device)
            self._engine().comm_destroy(nc)

    def use_gather(self, path):
        """Switch the field exchange of later reconstruct() calls ('auto' | 'p2p' | 'rccl'; COLLECTIVE like the calls
        themselves: every rank must switch at the same point).  A pending field is joined first."""
        self._flush_deferred()
        if path not in ('auto', 'p2p', 'rccl'):
            raise ValueError("path must be 'auto', 'p2p' or 'rccl'")
        pf = self.__dict__.get('_pending_field')
        if pf is not None and pf.pending:
            pf.wait()
        self._shard.gather = path
        self.__dict__.pop('_gather_sel', None)

    def _reconstruct_p2p(self, A_d, state, lay, to_host, wait):
        """reconstruct() over the p2p exchange: the kernel writes this rank's block straight into its persistent copy of
        the (n_p, n) field, the SDMA engines push the block into every peer's copy (any block sizes, no padding, no pass
        over the field afterwards).  The tensor returned is a view of that copy: valid until the next sharded
        reconstruct() of this object."""
        eng = self._engine()
        px = self._p2p
        Ur_d, rowmean_d, scale_d = state
        n_loc, n_p = Ur_d.shape[0], A_d.shape[0]
        first, total = int(lay[0, 0]), int(lay[:, 1].sum())
        pf = self.__dict__.get('_pending_field')
        if pf is not None and pf.pending:
            # an exchange nobody joined: its pushes still READ this rank's block of the copy the next kernel is about to
            # overwrite -- join it first (one single-wave kernel; every rank does the same, the call being collective)
            pf.wait()
        px.ensure(n_p, total)                                 # a larger field than before: new buffers (collective)
        if px.verified is None and px.peers and not px.loopback:
            return self._p2p_first_exchange(A_d, state, lay, to_host, wait)
        out = px.begin()
        off = self._row0 - first
        eng.reconstruct(Ur_d, self._row0, self.n_points, self.n_features, rowmean_d, scale_d, A_d,
                        out=out[:, off:off + n_loc])
        close = self._comm_bracket('gather')                  # issue -> join, when the join happens inside this call
        import time
        self.last_comm_ = ('field exchange (p2p)', (n_p, total), time.time())
        k = px.push(off, n_loc)
        if not to_host and not wait:
            pf = PendingField(out, join=lambda: px.join(k), needs_cus=False,
                              on_wait=lambda: self._comm_bracket('gather_exposed'))
            self._pending_field = pf
            return pf
        px.join(k)
        close()
        if not to_host:
            return out
        host = eng.to_host(out, result=True).T
        px.check()                                            # the host has just synchronised: did the join kernel give up?
        return host

    def _p2p_first_exchange(self, A_d, state, lay, to_host, wait):
        """The FIRST exchange through freshly mapped buffers (COLLECTIVE, once per allocation): the set-up's self-test moved
        32-byte patterns; this is the first time whole blocks cross the links, so the result is checked before anybody uses it.
        Every rank sums the bit patterns of its own block (int64, wrap-around: exact), the sums are all-gathered, and every rank
        compares them with the same sums over the blocks it RECEIVED; the join waits FIRST_TIMEOUT_S at most.  All ranks share
        the verdict.  On failure -- a HIP error in a push, a block that never arrives, a block that differs -- ``gather='auto'``
        drops to the collective all-gather for the rest of this object's life (``gather_path_`` says why) and this call returns
        that path's field; ``gather='p2p'`` raises on every rank.  On success ``gather='auto'`` goes on to time both exchanges
        (_gather_trial) and keeps the faster."""
        import os
        import sys
        eng = self._engine()
        t = eng.torch
        px = self._p2p
        Ur_d, rowmean_d, scale_d = state
        n_loc, n_p = Ur_d.shape[0], A_d.shape[0]
        first = int(lay[0, 0])
        off = self._row0 - first
        rank = self._shard.rank

        def block_sum(blk):
            return blk.view(t.int64).sum()

        ok, why, out = True, '', None
        mine = t.zeros((), dtype=t.int64, device=eng.device)
        keep = px.JOIN_TIMEOUT_S
        px.JOIN_TIMEOUT_S = min(keep, px.FIRST_TIMEOUT_S)
        try:
            out = px.begin()
            eng.reconstruct(Ur_d, self._row0, self.n_points, self.n_features, rowmean_d, scale_d, A_d,
                            out=out[:, off:off + n_loc])
            import time
            self.last_comm_ = ('field exchange (p2p, first: verified)', (n_p, int(lay[:, 1].sum())), time.time())
            k = px.push(off, n_loc)
            px.join(k)
            mine = block_sum(out[:, off:off + n_loc])
        except Exception as exc:                              # noqa: BLE001 -- any failure means "not this path", decided together
            ok, why = False, f'rank {rank}: {exc}'
        finally:
            px.JOIN_TIMEOUT_S = keep
        sums = eng.to_host(self._all_gather(mine.reshape(1))).reshape(-1)       # synchronises: the join kernel has ended
        if ok:
            try:
                px.check()
                got = eng.to_host(t.stack([block_sum(out[:, int(o) - first:int(o) - first + int(c)]) for o, c in lay]))
                bad = [int(q) for q in np.flatnonzero(got != sums)]
                if bad:
                    ok, why = False, f'rank {rank}: the blocks of ranks {bad} differ from what those ranks sent'
            except RuntimeError as exc:
                ok, why = False, str(exc)
        all_ok, why = px._agree(ok, why, 'the exchange failed')      # every rank's verdict and the first failing rank's reason
        want = os.environ.get('SPR_GATHER') or self._shard.gather
        if all_ok:
            px.verified = dict(blocks=int(lay.shape[0]), bytes_per_block=int(n_p * n_loc * 8), check='per-block int64 sums')
            if want == 'auto' and os.environ.get('SPR_GATHER_TRIAL', '1') != '0' and 'gather_trial_' not in self.__dict__:
                verdict = self._gather_trial(A_d, state)
                if verdict == 'rccl':
                    return self._reconstruct_now(A_d, state, to_host, wait)
                if verdict == 'p2p':
                    return self._reconstruct_p2p(A_d, state, lay, to_host, wait)    # (the trial's gathers reused the buffer)
                all_ok, why = False, verdict                  # the trial's p2p legs failed on some rank: as for a failed first exchange
            elif to_host:
                return eng.to_host(out, result=True).T
            else:
                return out if wait else PendingField(out)
        px.abandon()
        self._p2p_dropped = self.__dict__.pop('_p2p')          # stays allocated: peers have it mapped; never used again
        if want == 'p2p':
            self.__dict__.pop('_gather_sel', None)
            raise RuntimeError(f"RowShard(gather='p2p'): the first full-size exchange failed -- {why}")
        self._gather_sel = 'rccl'
        self.gather_path_ = f'rccl (p2p failed its first full-size exchange: {why})' if not str(why).startswith('failed: ') else f'rccl (p2p {why})'
        if rank == 0:
            print(f'[openmeasure_amd] field exchange falls back to the RCCL all-gather: {why}', file=sys.stderr)
        return self._reconstruct_now(A_d, state, to_host, wait)

    _GATHER_TRIAL_REPS = 2
    _GATHER_TRIAL_MARGIN = 0.97     # the all-gather takes over only when it is at least 3 % faster (p2p leaves the CUs alone)

    def _gather_trial(self, A_d, state):
        """``gather='auto'``, once per object, behind the verified first exchange (COLLECTIVE): WHICH exchange is faster HERE is
        a property of the node (links, SDMA engines, RCCL's protocol for this size) that nothing but a measurement can tell -- so
        the library measures what its callers do with a field exchange: leave it in flight under the next pass over X.  Per path,
        _GATHER_TRIAL_REPS times: reconstruct kernel + exchange enqueued unjoined, one Gram pass over the local rows queued behind
        it on the compute stream (the real kernel on the real shard, results discarded -- the stand-in for the next fit()), join,
        device sync; wall time.  A joined exchange alone would favour whichever path has the higher raw rate and miss that RCCL's
        kernel cannot share a compute unit with the Gram workgroups while the SDMA pushes do not need one.  Every rank takes the
        best of its repetitions, the maxima over the ranks decide (one all-gather: every rank sees the same two numbers), p2p
        keeps the exchange unless the all-gather is _GATHER_TRIAL_MARGIN faster.  ``gather_trial_`` / ``gather_path_`` carry both
        times.  SPR_GATHER_TRIAL=0 skips it (p2p whenever it is available, as before round 6).  -> 'p2p' | 'rccl'."""
        import sys
        import time
        eng = self._engine()
        torch = eng.torch
        Xd = self._Xd()
        can_fill = hasattr(eng, 'gram_filler')

        on_gpu = hasattr(eng, 'device') and eng.device.type == 'cuda'     # (the NumPy test double runs this on the CPU)

        def device_sync():
            if on_gpu:
                torch.cuda.synchronize(eng.device)

        def sync_all():
            self._all_gather(eng.zeros((1,)))                 # the ranks meet: every repetition starts together
            device_sync()

        def once(path):
            sync_all()
            t0 = time.perf_counter()
            pf = self._reconstruct_now(A_d, state, False, False, path=path)
            if can_fill:
                eng.gram_filler(Xd, Xd.shape[0], self._row0, self.n_points, self.n_features)
            if isinstance(pf, PendingField):
                pf.wait()
            device_sync()
            return time.perf_counter() - t0

        # HBM first: the all-gather leg stages the gathered field (world blocks) next to the block itself and RCCL allocates
        # buffers of its own at its first call of this size; a shard that fills the GPU (config 5 at N = 8: 9 GB left) must not
        # find out by running out of memory in one rank.  Decided together from the tightest rank.
        if on_gpu:
            lay = self._shard_layout(state[0].shape[0])
            need = (lay.shape[0] + 1) * int(lay[:, 1].max()) * A_d.shape[0] * 8 + (2 << 30)
            free = (torch.cuda.mem_get_info(eng.device)[0] + torch.cuda.memory_reserved(eng.device)
                    - torch.cuda.memory_allocated(eng.device))
            free = float(eng.to_host(self._all_gather(eng.to_device(np.array([float(free)])))).min())
            if free < need:
                self.gather_trial_ = dict(chosen='p2p', skipped=f'the all-gather leg needs about {need / 1e9:.1f} GB per rank, '
                                                                f'{free / 1e9:.1f} GB are left on the tightest rank')
                self.gather_path_ = ('p2p (SDMA pushes into peer-mapped buffers, no compute units; first-exchange trial skipped: '
                                     + self.gather_trial_['skipped'] + ')')
                return 'p2p'
        saved = self.comm_timing
        self.comm_timing = None                               # the trial's brackets are not the caller's
        px = self._p2p
        keep = px.JOIN_TIMEOUT_S
        px.JOIN_TIMEOUT_S = min(keep, px.FIRST_TIMEOUT_S)     # like the first exchange: nobody waits ten minutes inside a trial
        ok, why = True, ''
        best = {'p2p': np.inf, 'rccl': np.inf}
        try:
            once('rccl')                                      # the communicator's first all-gather of this size: not timed
            for _ in range(self._GATHER_TRIAL_REPS):
                for path in ('p2p', 'rccl'):
                    if path == 'p2p':
                        if not ok:                            # this rank's p2p leg has failed: it still meets the others ...
                            sync_all()
                            continue
                        try:
                            best[path] = min(best[path], once(path))
                        except Exception as exc:              # noqa: BLE001 -- ... and the verdict is agreed on below
                            ok, why = False, f'rank {self._shard.rank}: {exc}'
                    else:
                        best[path] = min(best[path], once(path))
            if ok:
                try:
                    px.check()                                # a join of the trial that gave up / read a poisoned counter
                except RuntimeError as exc:
                    ok, why = False, str(exc)
        finally:
            self.comm_timing = saved
            px.JOIN_TIMEOUT_S = keep
        all_ok, why = px._agree(ok, why, 'the p2p legs of the first-exchange trial failed')
        if not all_ok:
            self.gather_trial_ = dict(chosen='rccl', failed=why)
            return 'failed: ' + why
        mine = eng.to_device(np.array([best['p2p'], best['rccl']]))
        worst = eng.to_host(self._all_gather(mine)).max(axis=0)
        t_p2p, t_rccl = float(worst[0]), float(worst[1])
        sel = 'rccl' if t_rccl < self._GATHER_TRIAL_MARGIN * t_p2p else 'p2p'
        self.gather_trial_ = dict(p2p_ms=round(1e3 * t_p2p, 4), rccl_ms=round(1e3 * t_rccl, 4), chosen=sel,
                                  what='reconstruct + exchange left in flight under one Gram pass over the local rows, joined '
                                       'behind it; best of %d per rank, maximum over the ranks' % self._GATHER_TRIAL_REPS)
        said = f'exchange under a Gram pass {1e3 * t_p2p:.3f} ms over p2p, {1e3 * t_rccl:.3f} ms over the all-gather'
        if sel == 'p2p':
            self.gather_path_ = f'p2p (SDMA pushes into peer-mapped buffers, no compute units; first-exchange trial: {said})'
        else:
            self.gather_path_ = f'rccl (first-exchange trial: {said}; the p2p buffers stay mapped, use_gather("p2p") switches back)'
            if self._shard.rank == 0:
                print(f'[openmeasure_amd] field exchange: the RCCL all-gather is faster on this node ({said})', file=sys.stderr)
        self._gather_sel = sel
        return sel

    def _gather_unequal(self, A_d, state, lay, to_host, wait):
        """RCCL field all-gather for row blocks of different sizes: every rank contributes its (n_p, n_loc) block padded to
        the largest block (the all-gather wants equal counts; the reconstruct kernel writes into the padded block directly),
        and spr_field_unstage_blocks_f64 packs the blocks side by side afterwards -- one pass over the field, which the
        equal-shard path and the p2p exchange do not need."""
        import torch.distributed as dist
        eng = self._engine()
        Ur_d, rowmean_d, scale_d = state
        n_p, n_loc = A_d.shape[0], Ur_d.shape[0]
        world, n_max = lay.shape[0], int(lay[:, 1].max())
        mine = eng.zeros((n_p, n_max))
        eng.reconstruct(Ur_d, self._row0, self.n_points, self.n_features, rowmean_d, scale_d, A_d, out=mine[:, :n_loc])
        stage = eng.empty((world, n_p, n_max))
        close = self._comm_bracket('gather')
        nc = self._native_comm()
        if nc is not None:
            eng.comm_allgather(nc, mine, stage)
        else:
            dist.all_gather_into_tensor(stage.view(-1), mine.view(-1), group=self._shard.group)
        close()
        first = int(lay[0, 0])                                # 0 unless the group holds a slice of a larger job (partial)
        total = int(lay[:, 1].sum())
        table = np.stack([lay[:, 0] - first, lay[:, 1]], axis=1).astype(np.int64)
        if hasattr(eng, 'field_unstage_blocks'):
            out = eng.field_unstage_blocks(stage, table, total)
        else:                                                 # the NumPy test double
            out = eng.empty((n_p, total))
            for q in range(world):
                o, k = int(table[q, 0]), int(table[q, 1])
                out[:, o:o + k] = stage[q, :, :k]
        if to_host:
            return eng.to_host(out, result=True).T
        return out if wait else PendingField(out)
