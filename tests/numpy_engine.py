"""NumPy stand-in for openmeasure_amd.engine.HipEngine -- TEST DOUBLE, lives under tests/ only.

Same method set and tensor conventions as HipEngine, but on CPU torch tensors, so that the
host half of the SPR classes (validation, Gram-route algebra, Chan merges, sharding and the
collective call pattern over the gloo backend) can be exercised on machines without a GPU.
It models what the kernels compute (per-feature centred Gram, greedy residual-norm
pivoting with norm down-dating, normal equations + Cholesky), not how.  The package never
imports this file; without libspr_hip.so and a GPU the product raises instead.
"""
import numpy as np
import torch


class NumpyEngine:
    name = 'numpy-test-double'

    def __init__(self):
        self.torch = torch
        self.device = torch.device('cpu')

    # plumbing
    class _Tick:                                              # stand-in for a stream event: the host clock
        def __init__(self):
            import time
            self.t = time.perf_counter()

    def timing_event(self):
        return NumpyEngine._Tick()

    @staticmethod
    def elapsed_ms(e0, e1):
        return 1e3 * (e1.t - e0.t)

    def empty(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or torch.float64)

    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or torch.float64)

    def to_device(self, a, dtype=None):
        return torch.as_tensor(np.ascontiguousarray(a)).to(dtype or torch.float64).contiguous()

    def to_host(self, t, then=None, result=False):
        if then is not None:                                  # the HIP engine calls it between enqueueing the copy and blocking on it
            then()
        return t.detach().numpy()

    @staticmethod
    def _w(t):
        """float64 view of a stored matrix (f32 storage is widened, like the kernels do on load)"""
        return t.numpy().astype(np.float64, copy=False)

    @staticmethod
    def _feat(n, row0, n_points, F):
        return np.minimum((row0 + np.arange(n)) // n_points, F - 1)

    # K1 + K3a
    def stats_gram(self, X, row0, n_points, n_features, center=True):
        x = self._w(X)
        n, m = x.shape
        mean = x.mean(axis=1) if center else np.zeros(n)
        c = x - mean[:, None]
        feat = self._feat(n, row0, n_points, n_features)
        fstats = np.zeros((n_features, 3))
        gram = np.zeros((n_features, m, m))
        for f in range(n_features):
            sel = feat == f
            if sel.any():
                mu = mean[sel].mean()
                fstats[f] = (sel.sum(), mu, ((mean[sel] - mu) ** 2).sum())
                gram[f] = c[sel].T @ c[sel]
        return torch.from_numpy(mean.copy()), torch.from_numpy(fstats), torch.from_numpy(gram)

    # K4
    def project(self, X, row0, n_points, n_features, inv_scale, W, center=True, out=None, rowmean=None,
                basis_dtype=None, precenter=False):
        x = self._w(X)
        n = x.shape[0]
        mean = (rowmean.numpy() if rowmean is not None else x.mean(axis=1)) if center else np.zeros(n)
        feat = self._feat(n, row0, n_points, n_features)
        U = ((x - mean[:, None]) @ W.numpy()) * inv_scale.numpy()[feat][:, None]
        return torch.from_numpy(np.ascontiguousarray(U)).to(basis_dtype or torch.float64)   # f64 unless asked otherwise

    def project_f64(self, X, i0, rows, row0, n_points, n_features, inv_scale, W, rowmean, out, center=True,
                    precenter=False):
        x = self._w(X)[i0:i0 + rows]
        feat = self._feat(rows, row0 + i0, n_points, n_features)
        mean = rowmean.numpy()[i0:i0 + rows, None] if center else 0.0
        U = ((x - mean) @ W.numpy()) * inv_scale.numpy()[feat][:, None]
        out[:rows, :W.shape[1]] = torch.from_numpy(U)
        return out

    def feature_minmax(self, X, row0, n_points, n_features):
        x = self._w(X)
        feat = self._feat(x.shape[0], row0, n_points, n_features)
        out = np.empty((n_features, 2))
        for f in range(n_features):
            sel = feat == f
            out[f] = (x[sel].min(), x[sel].max()) if sel.any() else (np.inf, -np.inf)
        return torch.from_numpy(out)

    def feature_digit_hist(self, X, row0, n_points, n_features, prefix, shift, bits, two_targets):
        x = np.ascontiguousarray(self._w(X))
        feat = self._feat(x.shape[0], row0, n_points, n_features)
        u = x.view(np.uint64)
        sign = np.uint64(1) << np.uint64(63)
        key = np.where(u & sign, ~u, u | sign)
        pre = prefix.numpy().view(np.uint64)
        out = np.zeros((n_features, 2, 1 << bits), dtype=np.int64)
        top = shift + bits
        for f in range(n_features):
            k = key[feat == f].ravel()
            for t in range(2):
                sel = k if top >= 64 else k[(k >> np.uint64(top)) == (pre[f, t] >> np.uint64(top))]
                d = ((sel >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
                out[f, t] = np.bincount(d, minlength=1 << bits)
        return torch.from_numpy(out)

    def colsums(self, X, row0, n_points, n_features, rowmean):
        x, mu = self._w(X), rowmean.numpy()
        c = x - mu[:, None]
        feat = self._feat(x.shape[0], row0, n_points, n_features)
        out = np.zeros((n_features, 2, x.shape[1]))
        for f in range(n_features):
            sel = feat == f
            out[f, 0] = c[sel].sum(axis=0)
            out[f, 1] = (mu[sel, None] * c[sel]).sum(axis=0)
        return torch.from_numpy(out)

    def fill_feature(self, n_rows, row0, n_points, values):
        return torch.from_numpy(values.numpy()[self._feat(n_rows, row0, n_points, values.shape[0])].copy())

    def scale_rows(self, X, row0, n_points, n_features, rowmean, inv_scale):
        feat = self._feat(X.shape[0], row0, n_points, n_features)
        return torch.from_numpy((self._w(X) - rowmean.numpy()[:, None]) * inv_scale.numpy()[feat][:, None])

    def unscale(self, x0, row0, n_points, n_features, rowmean, scale, rowscale=None):
        feat = self._feat(x0.shape[0], row0, n_points, n_features)
        sc = scale.numpy()[feat] if rowscale is None else rowscale.numpy()
        return torch.from_numpy(sc * x0.numpy() + rowmean.numpy())

    # K10 + K11
    def reconstruct(self, Ur, row0, n_points, n_features, rowmean, scale, A, out=None, rowscale=None):
        feat = self._feat(Ur.shape[0], row0, n_points, n_features)
        sc = scale.numpy()[feat] if rowscale is None else rowscale.numpy()
        x = (self._w(Ur) @ A.numpy().T) * sc[:, None] + rowmean.numpy()[:, None]
        res = torch.from_numpy(np.ascontiguousarray(x.T))
        if out is not None:
            out.copy_(res)
            return out
        return res

    # K6
    def field_unstage(self, stage, out=None):
        world, n_p, n_loc = stage.shape
        return stage.permute(1, 0, 2).reshape(n_p, world * n_loc).contiguous()

    def mask_rows(self, Ur, mask_u8):
        Ur[mask_u8 == 0, :] = 0.0

    qr_batch = 8

    def _record(self, st):
        nrm = st['nrm'].numpy()
        i = int(np.argmax(nrm))                      # first index on ties
        second = np.partition(nrm, -2)[-2] if nrm.size > 1 else -2.0
        st['rec'] = torch.from_numpy(np.concatenate([[nrm[i], st['row0'] + i, second], self._w(st['Ur'])[i]]))

    def qr_begin(self, Ur, row0, n_steps):
        n, r = Ur.shape
        st = dict(Ur=Ur, n=n, r=r, row0=row0, nrm=torch.from_numpy((self._w(Ur) ** 2).sum(axis=1)),
                  Q=torch.zeros((n_steps, r), dtype=torch.float64),
                  piv=torch.zeros((n_steps,), dtype=torch.int64),
                  gap=torch.zeros((n_steps,), dtype=torch.float64),
                  ok=torch.zeros((n_steps,), dtype=torch.float64),
                  tau=torch.full((1,), -2.0, dtype=torch.float64))   # every row is a candidate here
        self._record(st)
        return st

    def qr_exclude(self, st, mask=None, xyz=None, n_points=1, j0=0, nq=0, d_min=0.0):
        nrm = st['nrm'].numpy()
        if mask is not None:
            nrm[mask.numpy() == 0] = -1.0
        if nq and xyz is not None:
            self._exclude_near(st, xyz.numpy(), n_points, st['piv'].numpy()[j0:j0 + nq], d_min)
        self._record(st)

    @staticmethod
    def _exclude_near(st, xyz, n_points, picks, d_min):
        nrm = st['nrm'].numpy()
        pos = xyz[(st['row0'] + np.arange(st['n'])) % n_points]
        for g in picks:
            if g >= 0:
                nrm[np.linalg.norm(pos - xyz[g % n_points], axis=1) < d_min] = -1.0

    def qr_step(self, st, step, recs, taus, first, xyz=None, n_points=0, d_min=0.0):
        c = recs.numpy()
        order = np.lexsort((c[:, 1], -c[:, 0]))      # max value, then lowest index
        w = order[0]
        piv = int(c[w, 1])
        st['piv'][step] = piv
        others = [c[w, 2]] + [c[i, 0] for i in range(c.shape[0]) if i != w]
        st['gap'][step] = (c[w, 0] - max(others)) / c[w, 0] if c[w, 0] > 0 else 0.0
        st['ok'][step] = 1.0 if (first or c[w, 0] > float(taus.max())) else 0.0
        v = c[w, 3:].copy()
        Q = st['Q'].numpy()
        for _ in range(2):
            v -= Q[:step].T @ (Q[:step] @ v)
        nn = np.linalg.norm(v)
        q = v / nn if nn > 0 else np.zeros_like(v)
        Q[step] = q
        nrm = st['nrm'].numpy()
        d = self._w(st['Ur']) @ q
        new = np.maximum(nrm - d * d, 0.0)
        new[nrm < 0] = -1.0
        li = piv - st['row0']
        if 0 <= li < st['n']:
            new[li] = -1.0
        nrm[:] = new
        if xyz is not None:
            self._exclude_near(st, xyz.numpy(), n_points, [piv], d_min)
        self._record(st)

    def qr_refresh(self, st, j0, nq):
        # the model down-dates every row at every step; only a direction that no step produced (GEM's
        # centring direction Q[0]) is applied here
        if st['piv'][j0] < 0:
            q = st['Q'].numpy()[j0]
            nrm = st['nrm'].numpy()
            d = self._w(st['Ur']) @ q
            new = np.maximum(nrm - d * d, 0.0)
            new[nrm < 0] = -1.0
            nrm[:] = new
            self._record(st)

    def qr_apply(self, st, dirs, picks):
        nrm = st['nrm'].numpy()
        U = self._w(st['Ur'])
        for q in dirs.numpy():
            d = U @ q
            new = np.maximum(nrm - d * d, 0.0)
            new[nrm < 0] = -1.0
            nrm[:] = new
        for g in picks.numpy():
            li = int(g) - st['row0']
            if g >= 0 and 0 <= li < st['n']:
                nrm[li] = -1.0
        self._record(st)

    # K7 + K8
    def measure_csr(self, indptr, indices, vals, Ur, row0, rowmean, scale=None, n_points=0):
        ip, ix, v = indptr.numpy(), indices.numpy(), vals.numpy()
        s, n, r = len(ip) - 1, Ur.shape[0], Ur.shape[1]
        Theta = np.zeros((s, r))
        cnt = np.zeros(s)
        scl = np.zeros(s)
        U, mu = self._w(Ur), rowmean.numpy()
        for i in range(s):
            for e in range(ip[i], ip[i + 1]):
                col = ix[e] - row0
                if 0 <= col < n:
                    Theta[i] += v[e] * U[col]
                    cnt[i] += v[e] * mu[col]
                    if scale is not None:
                        scl[i] += v[e] * scale.numpy()[min(ix[e] // n_points, scale.shape[0] - 1)]
        out = (torch.from_numpy(Theta), torch.from_numpy(cnt))
        return out if scale is None else out + (torch.from_numpy(scl),)

    # K8 + K9
    def solve_ols(self, Theta, cnt, scale, y):
        Th, c, sc, Y = Theta.numpy(), cnt.numpy(), scale.numpy(), y.numpy()
        n_p, s, r = Y.shape[0], Th.shape[0], Th.shape[1]
        Ar = np.zeros((n_p, r)); As = np.zeros((n_p, r)); y0 = np.zeros((n_p, s, 2)); info = np.zeros((n_p, 2))
        for p in range(n_p):
            scl = sc[Y[p, :, 2].astype(int)]
            y0[p, :, 0] = (Y[p, :, 0] - c) / scl
            y0[p, :, 1] = Y[p, :, 1] / scl
            weighted = np.any(Y[p, :, 1] != 0)
            w = 1.0 / y0[p, :, 1] if weighted else np.ones(s)
            A = Th * w[:, None]
            dn = np.sqrt(np.sum(A * A, axis=0))
            dn[dn == 0] = 1.0
            A = A / dn                                   # column equilibration, as the kernel does
            N = A.T @ A
            try:
                L = np.linalg.cholesky(N)
                d = np.diag(L)
                info[p, 1] = (d.max() / d.min()) ** 2
                Ar[p] = np.linalg.solve(N, A.T @ (w * y0[p, :, 0])) / dn
                if weighted:
                    As[p] = np.abs(np.linalg.solve(N, A.T @ y0[p, :, 1]) / dn)
            except np.linalg.LinAlgError:
                info[p, 0] = 1
        return (torch.from_numpy(Ar), torch.from_numpy(As), torch.from_numpy(y0), torch.from_numpy(info))

    def solve_pinv(self, Theta, cnt, scale, y, rcond=1e-15):
        Th, c, sc, Y = Theta.numpy(), cnt.numpy(), scale.numpy(), y.numpy()
        n_p, s, r = Y.shape[0], Th.shape[0], Th.shape[1]
        Ar = np.zeros((n_p, r)); As = np.zeros((n_p, r)); y0 = np.zeros((n_p, s, 2)); info = np.zeros((n_p, 4))
        for p in range(n_p):
            scl = sc[Y[p, :, 2].astype(int)]
            y0[p, :, 0] = (Y[p, :, 0] - c) / scl
            y0[p, :, 1] = Y[p, :, 1] / scl
            weighted = np.any(Y[p, :, 1] != 0)
            w = 1.0 / y0[p, :, 1] if weighted else np.ones(s)
            A = Th * w[:, None]
            sv = np.linalg.svd(A, compute_uv=False)
            Pi = np.linalg.pinv(A, rcond=rcond)
            Ar[p] = Pi @ (w * y0[p, :, 0])
            if weighted:
                As[p] = np.abs(Pi @ y0[p, :, 1])
            keep = sv > rcond * sv.max()
            info[p] = (1, keep.sum(), sv.max(), sv[keep].min())
        return (torch.from_numpy(Ar), torch.from_numpy(As), torch.from_numpy(y0), torch.from_numpy(info))



class CandidateEngine(NumpyEngine):
    """The pivoting half of the engine as a CANDIDATE-SET model (still a test double): sweeps leave, per block of
    ``block`` rows, the ``topt`` largest residual norms as candidates and tau = the largest norm a non-candidate may have;
    steps act on the candidates only and are certified against tau; refreshes, pool sweeps and full epoch sweeps redraw
    the candidates -- the protocol of csrc/qr_pivot.hip in NumPy, with blocks small enough that certification fails and
    pools empty out on matrices of a few hundred rows.  Lets the host drivers (pivot_loop, _pivot_loop_pooled, sharded or
    not) run on CPU; ``log`` records the refreshes taken."""

    qr_batch = 4

    def __init__(self, block=16, topt=2, max_dirs=8):
        super().__init__()
        self.block, self.topt, self.max_dirs = block, topt, max_dirs
        self.log = []

    # ---- candidate bookkeeping
    def _draw(self, st, visited, floor=-2.0):
        """candidates = per block the topt largest bounds among the visited rows; tau = the best bound left out"""
        nrm = st['nrm'].numpy()
        n = st['n']
        cand, tau = [], -2.0
        for b0 in range(0, n, self.block):
            rows = np.arange(b0, min(b0 + self.block, n))
            rows = rows[visited[rows]]
            if rows.size == 0:
                continue
            order = rows[np.lexsort((rows, -nrm[rows]))]            # value descending, lowest index first
            cand.extend(order[:self.topt].tolist())
            if order.size > self.topt:
                tau = max(tau, nrm[order[self.topt]])
        st['cand'] = np.array(cand, dtype=np.int64)
        st['cand_res'] = nrm[st['cand']].copy()
        st['tau'] = torch.tensor([max(tau, floor)], dtype=torch.float64)
        self._cand_record(st)

    def _cand_record(self, st):
        res, cand = st['cand_res'], st['cand']
        live = res >= 0
        U = self._w(st['Ur'])
        if not live.any():
            st['rec'] = torch.from_numpy(np.concatenate([[-2.0, 2.0 ** 62, -2.0], np.zeros(st['r'])]))
            return
        order = np.lexsort((cand, -res))
        order = order[live[order]]
        w = order[0]
        second = res[order[1]] if order.size > 1 else -2.0
        st['rec'] = torch.from_numpy(np.concatenate([[res[w], st['row0'] + cand[w], second], U[cand[w]]]))

    def qr_begin(self, Ur, row0, n_steps, norms=None):
        st = super().qr_begin(Ur, row0, n_steps)
        st['ldu'] = Ur.shape[1]
        if norms is not None:
            st['nrm'] = norms.clone()
        self._draw(st, np.ones(st['n'], dtype=bool))
        return st

    def qr_step(self, st, step, recs, taus, first, xyz=None, n_points=0, d_min=0.0):
        c = recs.numpy()
        order = np.lexsort((c[:, 1], -c[:, 0]))
        w = order[0]
        piv = int(c[w, 1])
        st['piv'][step] = piv
        others = [c[w, 2]] + [c[i, 0] for i in range(c.shape[0]) if i != w]
        st['gap'][step] = (c[w, 0] - max(others)) / c[w, 0] if c[w, 0] > 0 else 0.0
        st['ok'][step] = 1.0 if (first or c[w, 0] > float(taus.max())) else 0.0
        v = c[w, 3:].copy()
        Q = st['Q'].numpy()
        for _ in range(2):
            v -= Q[:step].T @ (Q[:step] @ v)
        nn = np.linalg.norm(v)
        Q[step] = v / nn if nn > 0 else 0.0
        d = self._w(st['Ur'])[st['cand']] @ Q[step]                 # the candidates only
        res = st['cand_res']
        new = np.maximum(res - d * d, 0.0)
        new[res < 0] = -1.0
        new[st['cand'] + st['row0'] == piv] = -1.0
        st['cand_res'] = new
        self._cand_record(st)

    def qr_steps(self, st, step0, n_steps, xyz=None, n_points=0, d_min=0.0, first_exact=True):
        for t in range(n_steps):
            self.qr_step(st, step0 + t, st['rec'][None], st['tau'][None], first=(t == 0 and first_exact))

    def _mark(self, st, arr, j0, j1):
        for g in st['piv'].numpy()[j0:j1]:
            li = int(g) - st['row0']
            if 0 <= li < st['n']:
                arr[li] = -1.0

    def _apply(self, st, base, rows, j0, j1):
        U = self._w(st['Ur'])[rows]
        d = U @ st['Q'].numpy()[j0:j1].T
        new = np.maximum(base[rows] - (d * d).sum(axis=1), 0.0)
        new[base[rows] < 0] = -1.0
        return new

    def qr_refresh(self, st, j0, nq):
        nrm = st['nrm'].numpy()
        self._mark(st, nrm, j0, j0 + nq)
        rows = np.arange(st['n'])
        nrm[:] = self._apply(st, nrm, rows, j0, j0 + nq)
        self.log.append(('refresh', j0, j0 + nq))
        self._draw(st, np.ones(st['n'], dtype=bool))

    # ---- epoch sweeps
    def qr_epoch_ok(self, st):
        return True

    def qr_epoch_max_directions(self, st):
        return self.max_dirs

    def qr_epoch_begin(self, st):
        st['nrm_e'] = st['nrm'].clone()
        st['pool_n'] = 0

    def qr_pool_build(self, st, theta):
        pool = np.flatnonzero(st['nrm_e'].numpy() > theta)
        if pool.size > max(st['n'] // 4, 8):
            st['pool_n'] = -1
            return -1
        st['pool'] = pool
        st['pool_n'] = int(pool.size)
        return st['pool_n']

    def qr_epoch_sweep(self, st, j_e, j, j_mark, pool=False, tau_floor=-2.0):
        assert 0 < j - j_e <= self.max_dirs, (j_e, j)
        nrm, nrm_e = st['nrm'].numpy(), st['nrm_e'].numpy()
        self._mark(st, nrm_e, j_mark, j)
        self._mark(st, nrm, j_mark, j)
        visited = np.zeros(st['n'], dtype=bool)
        if pool:
            rows = st['pool']
            nrm[rows] = self._apply(st, nrm_e, rows, j_e, j)
            visited[rows] = True
            self.log.append(('pool', j_e, j, int(rows.size)))
        else:
            rows = np.arange(st['n'])
            nrm[:] = self._apply(st, nrm_e, rows, j_e, j)
            nrm_e[:] = nrm
            visited[:] = True
            self.log.append(('full', j_e, j))
        self._draw(st, visited, tau_floor if pool else -2.0)


class ExchangeDoubleEngine(NumpyEngine):
    """NumpyEngine + a stand-in for the p2p field exchange (openmeasure_amd/p2p.py) -- TEST DOUBLE.  The product's sharded
    reconstruct drives an exchange object through ensure / begin / push / join / check (+ verified, abandon, close, _agree); the
    real one moves blocks with SDMA pushes into peer-mapped buffers and needs GPUs.  This one keeps the same interface and
    moves the blocks with a gloo all-gather at JOIN time, so that the host logic around the exchange -- path selection, the
    verified first exchange, the first-exchange trial and its agreement across ranks, the fall-back of all ranks together,
    deferred first launches -- runs on the CPU at any world size.  ``faults``: {'drop_after': k} makes THIS rank's pushes
    stop after its k-th (its peers then miss its block: every rank's next check() raises, as the join kernels' status words
    make the real one do)."""

    def __init__(self, faults=None):
        super().__init__()
        self.faults = dict(faults or {})
        self.exchanges = []

    def to_host(self, t, then=None, result=False):                    # a HOST COPY like the real engine's (the exchange buffer is
        return np.array(super().to_host(t, then=then, result=result))  # persistent: a view of it would change under the caller)

    def gram_filler(self, X, rows, row0, n_points, n_features):       # the trial queues a Gram pass behind the exchange
        self.filler_calls = getattr(self, 'filler_calls', 0) + 1

    def p2p_field_gather(self, world, rank, all_gather):
        from openmeasure_amd.p2p import P2PFieldGather

        eng = self

        class Exchange(P2PFieldGather):                      # inherits _agree (the ranks' verdicts and reasons, one all-gather)
            def __init__(self):                              # noqa: D401 -- none of the real set-up (library, streams, pinned words)
                self.eng, self.world, self.rank = eng, int(world), int(rank)
                self._all_gather = all_gather
                self.peers = [q for q in range(self.world) if q != self.rank]
                self.loopback, self.n_buf, self.k = 0, 1, 0
                self.base = self.buf = None
                self.shape, self.verified, self.memory = None, None, 'test double'
                self.host_ms = dict(begin=0.0, push=0.0, join=0.0, calls=0)
                self.pushed, self.joined, self.failed = {}, set(), ''
                self.pushes = 0

            def ensure(self, n_p, n_total):
                if self.buf is None or self.buf.shape[0] < n_p or self.buf.shape[1] != n_total:
                    self.buf = torch.zeros((int(n_p), int(n_total)), dtype=torch.float64)
                    self.base, self.k, self.verified = 1, 0, None
                self.shape = (int(n_p), int(n_total))

            def begin(self):
                self.check()
                return self.buf[:self.shape[0]]

            def push(self, first, n_loc):
                self.pushes += 1
                drop = eng.faults.get('drop_after')
                live = drop is None or self.pushes <= drop
                self.pushed[self.k] = (int(first), int(n_loc), live)
                k = self.k
                self.k += 1
                self.host_ms['calls'] += 1
                return k

            def join(self, k):
                """every rank's block of gather k into every rank's copy: one padded all-gather (collective at join time)"""
                if k in self.joined:
                    return
                self.joined.add(k)
                first, n_loc, live = self.pushed.pop(k)
                n_p, n_total = self.shape
                blk = torch.zeros((n_p, n_total + 3), dtype=torch.float64)
                blk[:, :n_loc] = self.buf[:n_p, first:first + n_loc]
                blk[:, n_total:] = torch.tensor([float(first), float(n_loc), 1.0 if live else 0.0], dtype=torch.float64)
                every = self._all_gather(blk)
                for q in range(self.world):
                    f, c, ok = (int(v) for v in every[q, 0, n_total:].tolist())
                    if not ok:
                        self.failed = self.failed or (f'rank {self.rank} gave up waiting for counter arrive[{q}] (the block of rank {q})'
                                                      if q != self.rank else f'my own pushes never left (rank {q})')
                    elif q != self.rank:
                        self.buf[:n_p, f:f + c] = every[q, :, :c]

            def check(self):
                if self.failed:
                    why, self.failed = self.failed, ''
                    raise RuntimeError('p2p field exchange (test double): ' + why)

            def abandon(self):
                self.failed = ''

            def close(self, collective=True, rendezvous=None):
                self.buf = self.base = None

        ex = Exchange()
        self.exchanges.append(ex)
        return ex
