"""Host drivers of the sensor placement (reference: scipy.linalg.qr(Ur.T, pivoting=True), sparse_sensing.py:739, and SPR.gem,
:586-698): the candidate-set pivot loop with epoch / pool sweeps over csrc/qr_pivot.hip, and the GEM methods of SPR (a mixin:
rom.SPR inherits them)."""
from __future__ import annotations

import numpy as np

_POOL_FRACTION = 1.0 / 16       # share of the rows a pool sweep visits
_POOL_USEFUL = 0.85             # a pool pays only if its threshold lies this far below the best row
_POOL_MARGIN = 1.15             # leave the pool when the winners have come this close to its threshold


def _pivot_loop_pooled(eng, st, s, stats, all_gather=None):
    """pivot_loop with epoch sweeps (csrc/qr_pivot.hip, qr_epoch_sweep_kernel): between two passes over the whole basis
    the refreshes visit only the POOL -- the rows whose norm at the start of the epoch lies above a threshold theta, about
    1/16 of them --, and the steps are certified against max(tau of the pool, theta): rows outside the pool keep their
    stale norms, which are upper bounds.  When the winners have come down to theta one full sweep starts the next epoch.
    Same pivots as a refresh per batch; at BASELINE config 3 two passes over Ur instead of four.
    Sharded runs: every rank keeps its own pool and decides on its own between a pool sweep and a full one -- the tau the
    ranks all-gather is each rank's own bound on its non-candidates, whatever refresh produced it.  Only the exactness of
    the first step after a refresh (which holds when EVERY rank has just swept all its rows) needs common knowledge: it
    is claimed after the initial norms and after a batch that certified nothing (all ranks then take a full sweep)."""
    torch = eng.torch
    n, batch = st['n'], eng.qr_batch
    dmax = eng.qr_epoch_max_directions(st)
    eng.qr_epoch_begin(st)
    stride = max(1, n // 65536)
    full, pooled = 1, 0

    def new_epoch():
        """threshold of the next pool from a sample of the epoch norms; -> theta or None (no pool)"""
        chk = eng.to_host(torch.cat([st['rec'][:1], st['nrm_e'][::stride]]))
        best, samp = chk[0], chk[1:]
        k = int(len(samp) * (1.0 - _POOL_FRACTION))
        theta = float(np.partition(samp, k)[k]) if 0 <= k < len(samp) else -1.0
        if not (0.0 < theta < _POOL_USEFUL * best):
            return None
        return theta if eng.qr_pool_build(st, theta) > 0 else None

    theta = new_epoch()
    j = j_e = j_mark = 0
    first_exact = True
    sharded = all_gather is not None
    while j < s:
        nb = min(batch, s - j)
        if not sharded:
            eng.qr_steps(st, j, nb, first_exact=first_exact)
        else:
            taus = all_gather(st['tau'])
            for t in range(nb):                                  # steps on the candidate set, no host sync
                eng.qr_step(st, j + t, all_gather(st['rec']), taus, first=(t == 0 and first_exact))
        if 'flat' in st:                                         # flags | record | tau in one buffer: one sync per batch, no cat
            chk = eng.to_host(st['flat'])
            n_all = st['ok'].shape[0]
            ok, best_next, tau = chk[j:j + nb], chk[n_all], chk[-1]
        else:
            chk = eng.to_host(torch.cat([st['ok'][j:j + nb], st['rec'][:1], st['tau']]))   # one sync per batch
            ok, best_next, tau = chk[:nb], chk[nb], chk[nb + 1]
        k = nb if ok.all() else int(np.argmin(ok))               # certified prefix
        if k < 1 and first_exact:
            raise RuntimeError('optimal_placement: first step after a sweep was not certified')
        # (k == 0 without first_exact: a tie between the best row and tau after a pool sweep -- or, sharded, after full
        #  sweeps nobody could vouch for; the full sweep below is then followed by a step that is exact by construction)
        j += k
        if j >= s:
            break
        # where the next winners are: the best remaining candidate after a fully certified batch, at most tau otherwise
        level = best_next if k == nb else tau
        use_pool = (theta is not None and k > 0 and level > _POOL_MARGIN * theta and j - j_e + batch <= dmax)
        if stats is not None:
            stats.setdefault('log', []).append(('batch', j - k, k, nb, float(level), float(tau), theta, st.get('pool_n', 0)))
        if use_pool:
            eng.qr_epoch_sweep(st, j_e, j, j_mark, pool=True, tau_floor=theta)
            pooled += 1
            first_exact = False
            if stats is not None:
                stats['log'].append(('pool sweep', j_e, j))
        else:
            while j - j_e > dmax:                                # more directions than one sweep applies (rare)
                eng.qr_epoch_sweep(st, j_e, j_e + dmax, j_mark)
                j_e += dmax
                j_mark = max(j_mark, j_e)
                full += 1
            eng.qr_epoch_sweep(st, j_e, j, j_mark)
            full += 1
            if stats is not None:
                stats['log'].append(('full sweep', j_e, j))
            j_e = j
            first_exact = (not sharded) or k == 0               # sharded: only k == 0 tells that every rank swept all rows
            theta = new_epoch() if s - j > batch // 2 else None
        j_mark = j
    if stats is not None:
        stats['pool_sweeps'] = pooled
    return full


def pivot_loop(eng, st, s, all_gather=None, start=0, near=None, pools=False, stats=None):
    """Host driver of the candidate-set pivoting (include/spr_hip.h, K6): batches of certified
    steps on the candidate set, one full sweep per batch.  Returns the number of sweeps over Ur.
    start: first step index (GEM keeps its centring direction in slot 0); near = (xyz, n_points, d_min):
    GEM's distance exclusion around every pick.  pools: refresh only the rows that can still be picked between two
    full sweeps (plain QR pivoting, bases the epoch-sweep kernel takes; see _pivot_loop_pooled)."""
    if (pools and near is None and start == 0 and s > eng.qr_batch
            and hasattr(eng, 'qr_epoch_ok') and eng.qr_epoch_ok(st)):
        return _pivot_loop_pooled(eng, st, s, stats, all_gather)
    kw = dict(xyz=near[0], n_points=near[1], d_min=near[2]) if near is not None else {}
    j, sweeps = start, 1
    while j < s:
        nb = min(eng.qr_batch, s - j)
        if all_gather is None and hasattr(eng, 'qr_steps'):   # one rank: the whole batch in one library call
            eng.qr_steps(st, j, nb, **kw)
        else:
            gather = all_gather if all_gather is not None else (lambda t: t[None])
            taus = gather(st['tau'])
            for t in range(nb):                               # steps on the candidate set, no host sync
                eng.qr_step(st, j + t, gather(st['rec']), taus, first=(t == 0), **kw)
        ok = eng.to_host(st['ok'][j:j + nb])                   # one sync per batch
        k = nb if ok.all() else int(np.argmin(ok))             # certified prefix (>= 1 by construction)
        if k < 1:
            raise RuntimeError('optimal_placement: first step after a sweep was not certified')
        j += k
        if j < s:
            if near is not None:
                eng.qr_exclude(st, j0=j - k, nq=k, **kw)
            eng.qr_refresh(st, j - k, k)
            sweeps += 1
    return sweeps


def _check_mask(mask, n_rows):
    """The search mask of optimal_placement is used as ``Ur[~mask, :] = 0`` (:737-738) / ``Ur[mask]`` (:621): a boolean vector of
    another length is NumPy's IndexError (its text); anything that is not a boolean vector has no meaning as a mask here (an integer
    array would index rows in the reference) and is refused."""
    if mask.dtype == np.bool_ and mask.ndim == 1 and mask.shape[0] != n_rows:
        raise IndexError(f'boolean index did not match indexed array along axis 0; size of axis is {n_rows} but size of '
                         f'corresponding boolean axis is {mask.shape[0]}')
    if mask.dtype != np.bool_ or mask.shape != (n_rows,):
        raise IndexError('mask must be a boolean array with one entry per (local) row')


class GemPlacement:
    """calc_type='gem' of SPR.optimal_placement and the public SPR.gem (mixed into rom.SPR)."""

    def gem(self, Ur, n_sensors, mask, d_min, verbose):
        """Reference :586-698, the method optimal_placement(calc_type='gem') calls with the fitted basis: greedy entropy
        placement on the rows of ``Ur`` (n_local, r) -> the ordered sensor rows (global indices).  ``Ur`` may be the fitted
        basis (``self.Ur``: the copy in HBM is used) or any other array of that many rows, which is uploaded for the call and
        leaves the fitted state as it was.  ``verbose``: the reference's table (:633-635, :652, :694) -- per sensor its row
        variance, its conditional variance given the earlier picks and the accumulated entropy, in the reference's scaled
        units -- recomputed on the host from the picked rows with the reference's own formulas (noise-free; _gem_table).

        The picks are the GEM sensors OF THE BASIS HANDED IN.  The row variances over the r entries (:622, :638) change when
        a column of Ur changes sign, and LAPACK's singular-vector signs are arbitrary (fit() here fixes them by its own
        rule, _sign_fix): after a plain fit() the sensors coincide with the reference's only if the bases coincide --
        fit(basis=(Ur_ref, Ar_ref)) -- not merely up to column signs."""
        self._flush_deferred()
        fitted = self._host.get('Ur')
        if Ur is fitted and 'Ur' in self._d:
            self._placement_gem(n_sensors, mask, d_min, verbose)
            return self.sensors_.copy()
        Ur = np.asarray(Ur)
        if Ur.ndim != 2 or Ur.shape[0] != self.X.shape[0]:
            raise ValueError(f'gem: Ur must have one row per (local) row of X, ({self.X.shape[0]}, r); got {Ur.shape}')
        eng = self._engine()
        t = eng.torch
        missing = object()
        keep_d = {k: (dict.get(self._d, k, missing), self._d.stash.get(k, missing)) for k in ('Ur', 'rowmean')}
        keep_r, keep_host = self.__dict__.get('r', missing), self._host.pop('Ur', missing)
        keep_attrs = {k: self.__dict__.get(k, missing) for k in ('_placed', 'sensors_', 'pivot_gap_', 'pivot_sweeps_')}
        try:
            self._d.stash.pop('Ur', None)
            self._d['Ur'] = eng.to_device(Ur, dtype=t.float32 if Ur.dtype == np.float32 else None)
            if 'rowmean' not in self._d:
                self._d['rowmean'] = eng.zeros((Ur.shape[0],))    # only its address is used (the measure kernel's centre output)
            self.r = int(Ur.shape[1])
            self._placement_gem(n_sensors, mask, d_min, verbose)
            return self.sensors_.copy()
        finally:
            for k, v in keep_attrs.items():                    # a placement on a foreign basis is not this object's placement
                if v is missing:
                    self.__dict__.pop(k, None)
                else:
                    self.__dict__[k] = v
            for k, (dev, host) in keep_d.items():
                dict.pop(self._d, k, None)
                if dev is not missing:
                    self._d[k] = dev
                if host is not missing:
                    self._d.stash[k] = host
            if keep_r is missing:
                self.__dict__.pop('r', None)
            else:
                self.r = keep_r
            if keep_host is not missing:
                self._host['Ur'] = keep_host

    def _gem_table(self, piv, file=None):
        """The table gem(verbose=True) prints in the reference (:633-635 header, :652 first row, :694 later rows): number of
        sensors, sigma^2 of the new sensor's row, its conditional variance given the earlier picks, accumulated entropy --
        all in the reference's scaled units (coef = 2 / sqrt(largest row variance), :622-624).  Recomputed on the host from
        the s picked rows of Ur with the reference's formulas (np.cov of the scaled picks, :660-678), without its unseeded
        noise; beyond r - 1 sensors, where the reference's value is decided by that noise, with the ridge this
        implementation puts in its place (_GEM_RIDGE).  The reference indexes its MASKED variance vector with the global row
        (:652, :694) -- right only without a mask; the variance printed here is the picked row's."""
        eng = self._engine()
        r = self.r
        piv = np.asarray(piv, dtype=np.int64)
        ip, ix, v = self._csr_device(None, (np.arange(len(piv) + 1), piv, np.ones(len(piv))))
        rows_d, _ = eng.measure_csr(ip, ix, v, self._d['Ur'], self._row0, self._d['rowmean'])
        Ua = np.asarray(eng.to_host(self._all_reduce(rows_d)), dtype=np.float64)
        coef = 2.0 / np.sqrt(np.var(Ua[0], ddof=1))           # the first pick is the row of largest variance (:641)
        A = Ua * coef
        sig = np.var(A, ddof=1, axis=1)
        header = ['# sensors', 'sigma^2 y', 'sigma^2 y|a', 'Htot']
        print(f"{'-'*70} \n {header[0]:^10} {header[1]:^10} {header[2]:^10} {header[3]:^10} \n ", file=file)
        H_tot = 0.0
        for s in range(len(piv)):
            if s == 0:
                print(f"{s+1:^10} {sig[s]:^10.2e} {'  -':^10} {'  -':^10}", file=file)
                continue
            Ac = A[:s] - A[:s].mean(axis=1, keepdims=True)
            S_aa = np.atleast_2d(Ac @ Ac.T / (r - 1))
            reg = self._GEM_RIDGE if s >= r - 1 else 0.0
            S_inv = 1.0 / S_aa if s == 1 else np.linalg.inv(S_aa + reg * np.eye(s))
            yc = A[s] - A[s].mean()
            S_ya = Ac @ yc / (r - 1)
            cond = float(yc @ yc / (r - 1) - S_ya @ S_inv @ S_ya)
            with np.errstate(invalid='ignore', divide='ignore'):
                H_tot += 0.5 * np.log(cond) + 0.5 * (np.log(2 * np.pi) + 1)
            print(f"{s+1:^10} {sig[s]:^10.2e} {cond:^10.2e} {H_tot:^10.2e}", file=file)

    def _placement_gem(self, n_sensors, mask, d_min, verbose=False):
        """calc_type='gem' (reference :586-698, :745-751): greedy maximisation of the conditional variance
        sigma_y^2 - S_ya S_aa^-1 S_ay of a row of Ur (its r entries as samples) given the rows picked so far,
        inside `mask`, never closer than d_min to an earlier pick.  That quantity is 1/(r-1) times the squared
        residual of the row, centred over its entries, after projecting out the centred picks, so the run is the
        QR pivoting above with the direction 1/sqrt(r) applied first.  The reference adds unseeded noise
        1e-5*N(0,1) to the diagonal of S_aa before inverting it (:667-668); this implementation is the
        noise-free limit, so picks whose lead over the runner-up is below that noise level are not comparable.
        Beyond r-1 sensors S_aa is singular and the reference's picks are decided by that noise alone."""
        eng = self._engine()
        Ur_d = self._fitted('Ur', 'Ur')
        r = self.r
        self._check_rank_cap("optimal_placement('gem')")
        # the reference loops `for s in range(n_sensors)` (:637): a non-integer is range()'s TypeError, zero or a negative count an
        # empty placement -- C of shape (0, n) (:748-749)
        if not isinstance(n_sensors, (int, np.integer)):
            raise TypeError(f"'{type(n_sensors).__name__}' object cannot be interpreted as an integer")
        n_sensors = int(n_sensors)
        if n_sensors < 1:
            self.sensors_ = np.zeros(0, dtype=np.int64)
            self.pivot_gap_ = np.zeros(0)
            self.pivot_sweeps_ = 0
            C = np.zeros((0, self._n_global))
            self._placed = (C, self.sensors_)
            return C
        if r < 3:
            raise NotImplementedError('gem needs at least three modes (row variances over r entries, r-1 >= 2 picks).')
        mask_d = None
        if mask is not None:
            mask = np.asarray(mask)
            _check_mask(mask, Ur_d.shape[0])
            mask_d = eng.to_device(mask.astype(np.uint8), dtype=eng.torch.uint8)
        near = None
        if d_min > 0:
            xyz = np.asarray(self.xyz, dtype=np.float64)
            if xyz.ndim != 2 or xyz.shape[0] != self.n_points or not 1 <= xyz.shape[1] <= 3:
                raise ValueError('gem with d_min > 0 needs xyz of shape (n_points, 1..3).')
            near = (eng.to_device(xyz), self.n_points, float(d_min))
        s = n_sensors
        s_exact = min(s, r - 1)                               # picks the noise-free rule defines (S_aa regular)
        st = eng.qr_begin(Ur_d, self._row0, s + 1)
        st['Q'][0] = r ** -0.5                                # centring direction; no row is attached to it
        st['piv'][0] = -1
        if mask_d is not None:
            eng.qr_exclude(st, mask=mask_d, n_points=self.n_points)
        eng.qr_refresh(st, 0, 1)
        gather = self._all_gather if self._dist() else None
        sweeps = 1 + pivot_loop(eng, st, s_exact + 1, gather, start=1, near=near)
        if s > s_exact:
            sweeps += self._gem_ridge_phase(st, s_exact, s, mask_d, near)
        self.pivot_sweeps_ = sweeps
        piv = eng.to_host(st['piv'])[1:].astype(np.int64)
        self.sensors_ = piv
        self.pivot_gap_ = eng.to_host(st['gap'])[1:]
        if verbose == True:                                   # noqa: E712  (the reference's comparison, :631)
            self._gem_table(piv)
        C = self._one_hot(piv, self._n_global)
        self._placed = (C, piv)
        return C

    _GEM_RIDGE = 1e-5      # RMS of the reference's unseeded regularisation noise (:667), in its scaled units

    def _gem_ridge_phase(self, st, s_exact, s, mask_d, near):
        """GEM picks beyond r-1 sensors.  After r-1 picks the covariance of the picked rows spans the whole centred
        space: every conditional variance is zero and the reference's choice is made by the unseeded noise
        1e-5*N(0,1) it adds to the diagonal of S_aa before inverting it (:667-668).  Documented deterministic
        stand-in: the noise is replaced by its RMS, S_aa + 1e-5 I (in the reference's scaled units, coef =
        2/sqrt(max row variance), :622-624), i.e. ridge-regularised conditional variances

            v(y) = (r-1) [s_yy - S_ya (S_aa + d I)^-1 S_ay] = d' u_y^T (U_a^T U_a + d' I)^-1 u_y,   d' = d (r-1),

        u = rows of Ur centred over their r entries.  v is what the sweep kernels' residual array holds: each pick u
        down-dates it by (g.u_y)^2 with the Sherman-Morrison direction g = A^-1 u sqrt(d'/(1 + u^T A^-1 u)),
        A = U_a^T U_a + d' I -- for d' -> 0 the orthonormalised residual direction of the noise-free phase.  A row
        is never picked twice; mask and d_min act as before.  One sweep over Ur per extra sensor."""
        eng = self._engine()
        t = eng.torch
        r = self.r
        Ur_d = self._d['Ur']
        piv = eng.to_host(st['piv'])[1:s_exact + 1].astype(np.int64)
        # the picked rows of Ur (each lives on one rank)
        ip, ix, v = self._csr_device(None, (np.arange(s_exact + 1), piv, np.ones(s_exact)))
        rows_d, _ = eng.measure_csr(ip, ix, v, Ur_d, self._row0, self._d['rowmean'])
        Ua = eng.to_host(self._all_reduce(rows_d))
        Uc = Ua - Ua.mean(axis=1, keepdims=True)
        var_max = np.sum(Uc[0] ** 2) / (r - 1)                 # the first pick is the row of largest variance (:641)
        dprime = self._GEM_RIDGE * var_max / 4.0 * (r - 1)
        # spectral form of A = Uc^T Uc + d' I INSIDE the centred space (orthogonal to 1): eigen-decompose P Uc^T Uc P with
        # P = I - 1 1^T / r and keep the r - 1 eigenvectors orthogonal to 1 -- also when the picked rows are numerically
        # dependent (duplicated rows of Ur, a mask / d_min that leaves fewer than r - 1 independent rows): the null
        # directions then simply carry lam = 0 and stay orthonormal instead of mixing with the 1-direction
        one = np.full(r, r ** -0.5)
        P = np.eye(r) - np.outer(one, one)
        lam, Qe = np.linalg.eigh(P @ (Uc.T @ Uc) @ P)
        along = np.abs(one @ Qe)                                # exactly one eigenvector lies along 1 (eigenvalue 0)
        keep = np.ones(r, dtype=bool)
        keep[int(np.argmax(along))] = False
        lam, Qe = np.maximum(lam[keep], 0.0), Qe[:, keep]
        Qe = Qe - np.outer(one, one @ Qe)
        Qe, _ = np.linalg.qr(Qe)                                # r - 1 orthonormal columns spanning the centred space
        lam = np.maximum(np.einsum('ij,ij->j', Qe, (Uc.T @ Uc) @ Qe), 0.0)
        Ainv = (Qe / (lam + dprime)) @ Qe.T
        # residual array from scratch: v = |u_c|^2 - sum_k lam_k/(lam_k + d') (q_k.u)^2
        st2 = eng.qr_begin(Ur_d, self._row0, s + 1)
        st2['piv'][:s_exact + 1] = st['piv'][:s_exact + 1]
        if mask_d is not None:
            eng.qr_exclude(st2, mask=mask_d, n_points=self.n_points)
        if near is not None:
            for j0 in range(1, s_exact + 1, eng.qr_batch):
                eng.qr_exclude(st2, xyz=near[0], n_points=near[1], j0=j0, nq=min(eng.qr_batch, s_exact + 1 - j0),
                               d_min=near[2])
        dirs = np.vstack([one[None, :], (Qe * np.sqrt(lam / (lam + dprime))).T])     # r directions: centring + r - 1
        # every picked row leaves the pool, whatever the number of directions: pad the pick list to the directions (and
        # apply any picks beyond them with zero directions)
        n_slots = max(dirs.shape[0], s_exact + 1)
        picks0 = np.full(n_slots, -1, dtype=np.int64)
        picks0[1:s_exact + 1] = piv
        if n_slots > dirs.shape[0]:
            dirs = np.vstack([dirs, np.zeros((n_slots - dirs.shape[0], r))])
        eng.qr_apply(st2, eng.to_device(dirs), eng.to_device(picks0, dtype=t.int64))
        sweeps = 1 + -(-dirs.shape[0] // eng.qr_batch)
        for j in range(s_exact + 1, s + 1):
            recs = eng.to_host(self._all_gather(st2['rec']))   # (ranks, r+3): value, global row, runner-up, row of Ur
            order = np.lexsort((recs[:, 1], -recs[:, 0]))
            best = recs[order[0]]
            if not best[0] > 0.0:
                raise RuntimeError('gem: no admissible row left for the remaining sensors (mask / d_min too strict)')
            others = [best[2]] + [recs[i, 0] for i in order[1:]]
            u = best[3:3 + r] - best[3:3 + r].mean()
            g = Ainv @ u
            den = 1.0 + u @ g
            Ainv -= np.outer(g, g) / den
            st2['piv'][j] = int(best[1])
            st2['gap'][j] = (best[0] - max(others)) / best[0]
            if near is not None:
                eng.qr_exclude(st2, xyz=near[0], n_points=near[1], j0=j, nq=1, d_min=near[2])
            eng.qr_apply(st2, eng.to_device((g * np.sqrt(dprime / den))[None, :]), st2['piv'][j:j + 1])
            sweeps += 1
        st['piv'][s_exact + 1:] = st2['piv'][s_exact + 1:]
        st['gap'][s_exact + 1:] = st2['gap'][s_exact + 1:]
        return sweeps
