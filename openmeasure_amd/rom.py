"""The class surface: ROM and SPR with the constructor, methods, defaults, return shapes and exception types of
``openmeasure.sparse_sensing`` (reference: src/openmeasure/sparse_sensing.py) over the engine -- see sparse_sensing.py for the
map of reference call sites to kernels.  Host eigen routes: _eigen.py; placement drivers and GEM: _placement.py; sharding,
collectives and the field exchange: _shard.py."""
from __future__ import annotations

import numpy as np

from . import _eigen
from ._lib import SPR_MAX_R_WIDE
from ._placement import GemPlacement, _check_mask, pivot_loop
from ._shard import PendingField, RowShard, ShardedOps

__all__ = ['ROM', 'SPR', 'RowShard', 'DeviceMatrix', 'PendingField', 'OneHotRows']

_DEVICE_SPECTRUM_MAX_M = 24   # above this the single-workgroup Jacobi is slower than host dsyevd (csrc/spectrum.hip)
# Gram route (fit): singular vectors come from the eigenvectors of X0^T X0, whose rounding error eps * sigma_1^2 reaches
# mode i as eps * (sigma_1/sigma_i)^2.  Up to this ratio the basis is as good as LAPACK's to ~1e-8 (sensor parity
# verified on reference fixtures); above it fit() runs the refinement pass of _refine_spectrum (one more read of X).
_GRAM_KAPPA_REFINE = 1e4
_GRAM_REFINE_MAX_PASSES = 3
_DENSE_C_LIMIT = 1 << 26   # optimal_placement returns a dense ndarray below this many bytes (64 MiB)


class DeviceMatrix:
    """A snapshot block that already lives in HBM (2-D float64 or float32 CUDA tensor, rows contiguous).
    float32 = storage precision only: every kernel widens on load and computes in f64, and the basis Ur is float64
    like the reference's (its X0 = (X - X_cnt)/X_scl is float64 for any dtype of X because X_cnt / X_scl are, :106-107,
    :169, and np.linalg.svd :272 keeps that dtype).  ``basis='f32'`` stores the basis in float32 as well -- a storage
    option the reference does not have, for shards whose f64 basis would not fit (BASELINE config 5: 204.8 GB shard +
    51.2 GB f32 basis per GPU); the sensors are then those of the stored basis and ``pivot_gap_`` says how far each
    pick was from a tie."""

    def __init__(self, tensor, basis=None):
        if basis not in (None, 'f64', 'f32'):
            raise ValueError("basis must be None / 'f64' / 'f32'")
        self.tensor = tensor
        self.basis = 'f32' if basis == 'f32' else 'f64'
        if self.basis == 'f32' and str(tensor.dtype) != 'torch.float32':
            raise ValueError("basis='f32' is a storage option of a float32 snapshot block")

    @property
    def shape(self):
        return tuple(self.tensor.shape)


class OneHotRows:
    """The one-hot measurement matrix C of ``optimal_placement`` (reference :741-743: ``C = np.zeros((s, n));
    C[j, P[j]] = 1``) held as its s row indices instead of s x n doubles -- 46 GB at BASELINE config 3, 819 GB at
    config 5.  ``optimal_placement`` returns the plain ndarray while it is small (64 MiB) and this object above that;
    it behaves like the dense matrix for everything the reference's documentation does with C (README.md:165-184):

        np.argmax(C[i, :])          C @ x, C.dot(x)  (x of shape (n,) or (n, k))        C.shape, len(C)
        np.argmax(C, axis=1)        spr.train(C)                                         np.asarray(C), C.toarray()

    ``np.asarray`` / ``toarray`` build the dense matrix only below ``dense_limit`` bytes (MemoryError beyond: that
    allocation is what this class exists to avoid); ``tocsr()`` gives a scipy.sparse matrix of any size.  ``rows`` are
    the ordered global sensor rows (= ``spr.sensors_``).  A 1-D instance (shape (n,)) is what ``C[i, :]`` returns."""

    dense_limit = 1 << 32
    dtype = np.dtype(np.float64)
    __array_priority__ = 20.0                                  # ndarray @ OneHotRows defers to __rmatmul__

    def __init__(self, rows, n, _vector=False):
        self.rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        self.n = int(n)
        self._vector = bool(_vector)
        if self._vector and self.rows.size != 1:
            raise ValueError('a one-hot vector has exactly one non-zero')
        if self.rows.size and (self.rows.min() < 0 or self.rows.max() >= self.n):
            raise IndexError('sensor row outside [0, n)')

    @property
    def shape(self):
        return (self.n,) if self._vector else (self.rows.size, self.n)

    @property
    def ndim(self):
        return 1 if self._vector else 2

    @property
    def nnz(self):
        return int(self.rows.size)

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f'OneHotRows(shape={self.shape}, rows={self.rows.tolist() if self.rows.size <= 16 else "..."})'

    # ---- what NumPy asks for -------------------------------------------------------------------------------
    def argmax(self, axis=None, out=None, **kw):
        """np.argmax(C[i, :]) -> the sensor's row; np.argmax(C, axis=1) -> all of them (1-D); axis=None: flat index."""
        if self._vector:
            if axis not in (None, 0, -1):
                raise np.exceptions.AxisError(axis, 1)
            return np.int64(self.rows[0])
        if axis in (1, -1):
            return self.rows.copy()
        if axis is None:                                       # first maximum of the flattened matrix: row 0's one
            return np.int64(self.rows[0]) if self.rows.size else np.int64(0)
        if axis in (0, -2):
            raise MemoryError('argmax over axis 0 of the one-hot matrix needs its n columns; use tocsr()')
        raise np.exceptions.AxisError(axis, 2)

    def __array__(self, dtype=None, copy=None):
        nbytes = 8 * self.n * (1 if self._vector else self.rows.size)
        if nbytes > self.dense_limit:
            raise MemoryError(f'dense form of this one-hot matrix needs {nbytes / 2 ** 30:.1f} GiB; use its row indices '
                              '(.rows / np.argmax(C, axis=1)), C @ x, or C.tocsr()')
        if self._vector:
            out = np.zeros(self.n)
            out[self.rows[0]] = 1.0
        else:
            out = np.zeros((self.rows.size, self.n))
            out[np.arange(self.rows.size), self.rows] = 1.0
        return out if dtype is None else out.astype(dtype, copy=False)

    def toarray(self):
        return self.__array__()

    def tocsr(self):
        import scipy.sparse as sp
        s = self.rows.size
        return sp.csr_matrix((np.ones(s), self.rows, np.arange(s + 1)), shape=(s, self.n))

    def sum(self, axis=None):
        if self._vector or axis is None:
            return np.float64(self.rows.size)
        if axis in (1, -1):
            return np.ones(self.rows.size)
        return np.bincount(self.rows, minlength=self.n).astype(np.float64)   # axis 0: (n,), allocated by request

    # ---- indexing: C[i, :], C[i], C[i, j], C[a:b] ----------------------------------------------------------
    def __getitem__(self, key):
        if self._vector:
            if isinstance(key, (int, np.integer)):
                k = int(key) + (self.n if key < 0 else 0)
                if not 0 <= k < self.n:
                    raise IndexError('index out of range')
                return np.float64(1.0 if k == self.rows[0] else 0.0)
            return np.asarray(self)[key]
        if isinstance(key, tuple):
            if len(key) != 2:
                raise IndexError('too many indices for a 2-D matrix')
            ri, ci = key
        else:
            ri, ci = key, slice(None)
        full_cols = isinstance(ci, slice) and ci == slice(None)
        if isinstance(ri, (int, np.integer)):
            row = self.rows[ri]                                # IndexError like ndarray when out of range
            if full_cols:
                return OneHotRows([row], self.n, _vector=True)
            if isinstance(ci, (int, np.integer)):
                k = int(ci) + (self.n if ci < 0 else 0)
                if not 0 <= k < self.n:                        # the dense matrix this stands in for raises as well
                    raise IndexError(f'index {int(ci)} is out of bounds for axis 1 with size {self.n}')
                return np.float64(1.0 if k == row else 0.0)
            return np.asarray(OneHotRows([row], self.n, _vector=True))[ci]
        if full_cols:
            return OneHotRows(self.rows[ri], self.n)           # row slice / index array: still one-hot rows
        return np.asarray(self)[key]                            # column subsets: dense (size-guarded)

    # ---- products ------------------------------------------------------------------------------------------
    def dot(self, x):
        """C.dot(x) = the sampled entries x[rows] (reference :797, :573 and README.md:176)."""
        x = np.asarray(x)
        if x.ndim == 0 or x.shape[0] != self.n:
            raise ValueError(f'shapes {self.shape} and {x.shape} not aligned')
        return x[self.rows[0]] if self._vector else x[self.rows]

    __matmul__ = dot

    def __rmatmul__(self, a):
        return np.asarray(a) @ np.asarray(self)                 # (k, s) @ (s, n): dense by nature, size-guarded

    @property
    def T(self):
        return self.tocsr().T


class _Trace:
    """Optional per-phase wall-clock trace of fit() (SPR_TRACE=1): synchronises at every mark."""

    def __init__(self, eng):
        import os
        self.on = os.environ.get('SPR_TRACE', '0') == '1'
        self.eng = eng
        self.marks = []
        if self.on:
            self.mark('start')

    def mark(self, label):
        if not self.on:
            return
        import time
        if hasattr(self.eng, 'device') and self.eng.device.type == 'cuda':
            self.eng.torch.cuda.synchronize()
        self.marks.append((label, time.perf_counter()))

    def report(self):
        if not self.on:
            return
        import sys
        t0 = self.marks[0][1]
        prev = t0
        parts = []
        for label, t in self.marks[1:]:
            parts.append(f'{label}={1e3 * (t - prev):.2f}')
            prev = t
        print('[spr trace] ' + ' '.join(parts) + f' total={1e3 * (prev - t0):.2f} ms', file=sys.stderr)


class _DeviceState(dict):
    """The tensors an object keeps in HBM, by name.  An unpickled object starts with host copies only (``stash``): an
    entry is uploaded when it is first asked for, so loading a pickle needs no GPU until the object is used."""

    def __init__(self, stash=None, upload=None):
        super().__init__()
        self.stash = dict(stash or {})
        self.upload = upload

    def __missing__(self, key):
        if key in self.stash:
            t = self[key] = self.upload(self.stash.pop(key))
            return t
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.stash

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def pop(self, key, *default):
        if not dict.__contains__(self, key) and key in self.stash:
            # every pop in this module either discards the entry or wants a BUFFER to overwrite: a host copy is neither
            del self.stash[key]
            if default:
                return default[0]
            raise KeyError(key)
        return dict.pop(self, key, *default)


#: attributes that never travel in a pickle: the engine, device state (downloaded instead), events, in-flight work
_TRANSIENT = ('_eng', '_d', '_trace', '_pending', '_pending_field', '_gram_events', '_gram_events_pending', '_layout_src',
              '_gap_t0', '_G', '_last_decomp', 'comm_timing', 'last_comm_', '_layout', '_p2p', '_gather_sel', '_basis_checked', '_basis_diverged', '_deferred', '_row0_d', '_ncomm', '_combined', '_comm_stream')


class ROM(ShardedOps):
    """Reduced-order-model utilities (reference: ROM, sparse_sensing.py:18-511)."""

    def __init__(self, X, n_features, xyz, shard=None, engine=None):
        # reference :69-81 -- same checks, same exception types, in the same order
        if isinstance(X, DeviceMatrix):
            pass
        elif type(X) is not np.ndarray:
            raise TypeError('The matrix X is not a numpy array.')
        if type(n_features) is not int:
            raise TypeError('The parameter n_features is not an integer.')
        self.X = X
        self.n_features = n_features
        self.xyz = xyz
        self._shard = shard
        n = shard.n_global if shard is not None else X.shape[0]
        self.n_points = n // self.n_features
        if n % self.n_features != 0:
            raise Exception('The number of rows of X is not a multiple of n_features')
        self._n_global = n
        self._row0 = shard.row0 if shard is not None else 0
        if self._row0 + X.shape[0] > n:
            raise ValueError('The local row block does not fit in the global matrix.')
        self._eng = engine
        self._d = _DeviceState()    # device-resident state
        self._host = {}         # lazily downloaded copies

    # ------------------------------------------------------------------ reference methods outside the built path
    def CPOD(self, problem_dict, **kwargs):
        """Reference :434-461: the constrained POD solves one cvxpy problem per snapshot.  Not built (no conic solver on
        the device, cvxpy not available to pin a result against): raises like every option without a device path."""
        raise NotImplementedError('CPOD (constrained POD through cvxpy, reference :434-461) is not part of this implementation.')

    def adaptive_sampling(self, P, scale_type='std'):
        """Reference :377-432.  Not built: its snapshot weights contain Vt[k,:] @ V[k,:] (:401), which changes with the
        arbitrary signs LAPACK gives the singular vectors, and its candidate points come from an unseeded Latin hypercube --
        there is no result to reproduce."""
        raise NotImplementedError('adaptive_sampling (reference :377-432) is not part of this implementation: its result '
                                  'depends on the sign convention of the singular vectors and on an unseeded sampler.')

    # ------------------------------------------------------------------ pickling (the reference's objects are plain attributes)
    def __getstate__(self):
        """What the reference's object would pickle -- X, the fitted arrays, the trained operator -- with everything that
        lives in HBM downloaded first (a DeviceMatrix becomes the host ndarray of its values; the basis comes down whole: a
        pickle of a fitted config-3 object is 230 GB, as the reference's would be).  The engine, events and in-flight
        work stay behind; a process group cannot travel either (the shard keeps its row block, group = default)."""
        self._flush_deferred()
        self._materialize()
        pf = self.__dict__.get('_pending_field')
        if pf is not None and pf.pending:
            pf.wait()
        eng = self._eng
        state = {k: v for k, v in self.__dict__.items() if k not in _TRANSIENT and not hasattr(v, 'is_cuda')}
        d_host = dict(self._d.stash)
        for k, t in self._d.items():
            if k == 'X' and not isinstance(self.X, DeviceMatrix):
                continue                                       # the host ndarray X is in the state already
            d_host[k] = eng.to_host(t)
        if isinstance(self.X, DeviceMatrix):
            state['X'] = d_host.pop('X') if 'X' in d_host else eng.to_host(self.X.tensor)
            state['_basis_f32'] = self.X.basis == 'f32'
        state['_d_host'] = d_host
        if self._shard is not None:
            sh = RowShard(self._shard.row0, self._shard.n_global, None, self._shard.force_collectives,
                          self._shard.broadcast_basis, self._shard.partial, self._shard.gather, self._shard.native_comm)
            state['_shard'] = sh
        return state

    def __setstate__(self, state):
        d_host = state.pop('_d_host', {})
        self.__dict__.update(state)
        self._eng = None                                       # the default engine, created on first use (or assign ._eng)

        def upload(a):
            eng = self._engine()
            return eng.to_device(a, dtype=eng.torch.float32 if a.dtype == np.float32 else None)
        self._d = _DeviceState(d_host, upload)

    # ------------------------------------------------------------------ lazily fetched fit results
    _LAZY = ('Ar', 'Sigma_r', 'Vr', 'exp_variance_', 'S_', '_scl_f', '_var_f')

    #: Squared row norms of the basis as a by-product of fit(): the projection that stores Ur also leaves |Ur[i, :]|^2 of
    #: every row it wrote (8 bytes per row), and optimal_placement('qr') starts from that vector instead of reading the
    #: whole basis once more (one sweep of 46 GB less at BASELINE config 3; same sensors).  None ("auto", the default of
    #: SPR): whenever the projection kernel the shape takes anyway produces them at no measurable cost (the W-stationary
    #: and the streamed-W kernel: every BASELINE shape but config 1); True: always (m <= 256 shapes outside the
    #: W-stationary kernel's range then run the streamed-W kernel); False (ROM, whose users never place sensors): never.
    placement_norms = False

    #: Gap filler (OPT-IN: ``rom.gap_filler = True`` or SPR_GAP_FILLER=1).  Between the Gram pass and the projection the
    #: device waits for the host (download of the m x m Gram matrix, eigen-solve, upload of W: 1.8 ms at m = 256, 8 ms at
    #: m = 512), and a chip that idles for a few milliseconds lowers its clock and takes several more to raise it again: the
    #: kernels of a fit() + reconstruct() step on one rank's block of BASELINE config 4 at N = 8 take 19.9 ms behind a 3 ms gap
    #: against 18.6 ms back to back (tools/keepalive_probe.py).  With the filler on, fit() queues the Gram kernel ONCE MORE
    #: over the first rows of X, sized to 85 % of the shortest host gap of the last fits, before it blocks on the download;
    #: the results are discarded.  It costs energy (about 2.5 J per fit) and, on a GPU shared with other work, their time --
    #: which is why a library must not do it unasked.  Only for m >= 128 and host gaps of at least _GAP_FILL_MIN_MS (shorter
    #: ones leave no clock drop worth filling).  Sharded objects size it from their own host-gap history: ranks with different
    #: histories queue fillers of different length in front of their projections (unmeasured on more than one GPU).
    gap_filler = False

    #: Deferred reconstruct (default since round 6; ``rom.defer_reconstruct = False`` or SPR_DEFER_RECONSTRUCT=0 switch it off).
    #: Between the Gram pass and the projection of fit() the device idles while the host eigen-solves (0.3 ms of a 1.5 ms step at
    #: BASELINE config 2, 1.8 ms of 22 on one rank's block of config 4 at N = 8).  ``reconstruct(a, to_host=False, wait=False)``
    #: -- the asynchronous form, which promises the field only behind ``PendingField.wait()`` -- therefore does not launch: the
    #: kernel (and, sharded, the push of the block to the peers) is enqueued in the host gap of the object's NEXT fit(), before
    #: that fit's projection overwrites the basis (the launch holds the basis, centre and scale tensors of the fit it was called
    #: after), or by ``wait()`` / any other method of the object, whichever comes first.  Same results, bit for bit
    #: (tools/step_stress.py, tools/p2p_stress.py run with it on); a loop fit -> reconstruct -> fit -> ... fills its gaps with
    #: useful work: config 2 1.49-1.52 -> 1.27-1.31 ms per step, one rank's block of config 4 22.3 -> 20.8-21.3 ms, config 3
    #: 158.1 -> 155.6 (profiles/r05_defer_reconstruct_ab.txt).  Every other form of reconstruct() -- the reference's host array,
    #: ``wait=True`` -- launches at once, as before.
    defer_reconstruct = True

    def _flush_deferred(self):
        """Launch the reconstruct a previous reconstruct(wait=False) deferred (defer_reconstruct); -> True if there was one"""
        pf = self.__dict__.pop('_deferred', None)
        if pf is not None and not pf.launched:
            pf.launch()
            return True
        return False
    _GAP_FILL_FRACTION = 0.85
    _GAP_FILL_MIN_M = 128
    #: ... and only when the host gap is long enough for the clock to matter: behind a 2.8 ms gap the filler took 0.7 ms off a
    #: 23 ms step, behind the 1.8 ms gap the batched eigen-solve leaves at m = 256 it changes nothing (same-box A/B, three
    #: repetitions: 21.98-22.37 vs 22.09-22.13 ms).  Hosts whose eigen-solve takes longer (other CPUs, loaded machines) fill, and so
    #: do wide matrices: behind the 7.6 ms eigen-solve of m = 512 the filler takes 1.6 ms off a 35 ms projection (c5s 114.0 -> 112.4 ms).
    _GAP_FILL_MIN_MS = 2.2

    #: Collective timing (bench.py): set to a dict and every collective of fit() / reconstruct() appends a pair of
    #: engine timing events (recorded on the stream the collective is ordered on) under 'allreduce' (the ONE all-reduce
    #: of fit), 'gather' (reconstruct's field all-gather when it is joined inside the call) or 'gather_exposed' (the
    #: join of a gather that was left in flight, PendingField.wait(): what of it did NOT hide under the next pass).
    comm_timing = None

    def _comm_bracket(self, key):
        """-> a function that closes the bracket opened now (no-op when comm_timing is off)."""
        ct = self.comm_timing
        eng = self._engine()
        if ct is None or not hasattr(eng, 'timing_event'):
            return lambda: None
        e0 = eng.timing_event()

        def close():
            ct.setdefault(key, []).append((e0, eng.timing_event()))
        return close

    def __getattr__(self, name):
        # only reached when normal lookup fails: results of the sync-free device fit (m <= 64) stay in HBM
        # until somebody reads them
        pend = self.__dict__.get('_pending')
        if pend is not None and name in ROM._LAZY:
            self._materialize()
            if name in self.__dict__:
                return self.__dict__[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def _materialize(self):
        """One D2H round trip for everything the device spectrum left in HBM."""
        pend = self.__dict__.pop('_pending', None)
        if pend is None:
            return
        eng = self._engine()
        r = pend['r']
        feat = eng.to_host(pend['feat'])
        S = eng.to_host(pend['S'])
        V = eng.to_host(pend['V'])
        self._scl_f = feat[:, 3].copy()
        self._var_f = self._scl_f ** 2
        self.S_ = S
        self.exp_variance_ = eng.to_host(pend['expvar'])[:r].copy()
        self.Ar = eng.to_host(pend['Ar'])
        self.Sigma_r = np.linalg.norm(self.Ar, axis=0)         # :504-508
        Vr = V[:, :r]
        self.Vr = Vr * (np.linalg.norm(Vr, axis=0) ** -1)

    # ------------------------------------------------------------------ plumbing
    def _engine(self):
        if self._eng is None:
            from .engine import HipEngine
            self._eng = HipEngine()
        return self._eng

    def _Xd(self):
        if 'X' not in self._d:
            eng = self._engine()
            if isinstance(self.X, DeviceMatrix):
                self._d['X'] = self.X.tensor
            else:
                if self.X.ndim < 2:
                    # the reference's first touch of X's columns is x = X[i n_points:(i+1) n_points, :] (:110): NumPy's text
                    raise IndexError(f'too many indices for array: array is {self.X.ndim}-dimensional, but 2 were indexed')
                if self.X.ndim != 2:
                    # ... and for more dimensions np.average(x, axis=1) no longer fits the n_points rows it is assigned to (:112)
                    shp = ','.join(str(d) for d in (self.n_points,) + tuple(self.X.shape[2:]))
                    raise ValueError(f'could not broadcast input array from shape ({shp}) into shape ({self.n_points},)')
                # a float32 snapshot matrix stays float32 in HBM (storage only, see DeviceMatrix)
                self._d['X'] = eng.to_device(self.X, dtype=eng.torch.float32 if self.X.dtype == np.float32 else None)
        return self._d['X']

    def _norms_buffer(self, Xd, r, precenter=False):
        """The vector the projection writes the squared row norms to (see placement_norms), or None."""
        eng = self._engine()
        want = self.placement_norms
        if want is False or r > 128 or not getattr(eng, 'supports_row_norms', False):
            return None
        if want is None and not eng.project_writes_norms(Xd, r, True, precenter):
            return None
        return eng.empty((Xd.shape[0],))

    def _basis_dtype(self):
        """storage type of Ur: float64 (the reference's, for any dtype of X) unless DeviceMatrix(basis='f32')"""
        t = self._engine().torch
        f32 = (isinstance(self.X, DeviceMatrix) and self.X.basis == 'f32') or self.__dict__.get('_basis_f32', False)
        return t.float32 if f32 else t.float64

    def _lazy(self, key, make):
        if key not in self._host:
            self._host[key] = make()
        return self._host[key]

    def _fitted(self, key, attr):
        """Device-resident state `key`; before fit()/scale_data() the reference fails with AttributeError on `attr`."""
        try:
            return self._d[key]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'") from None

    def _feature_rows(self):
        """per local row: its feature id (host, int) -- only used to expand per-feature scalars"""
        n_loc = self.X.shape[0]
        return (self._row0 + np.arange(n_loc)) // self.n_points

    # ------------------------------------------------------------------ fitted attributes
    @property
    def X_cnt(self):
        """(n_local, 1) row means (reference attribute set at :166)."""
        return self._lazy('X_cnt', lambda: self._engine().to_host(self._fitted('rowmean', 'X_cnt'))[:, None])

    @property
    def X_scl(self):
        """(n_local, 1) per-feature population std, repeated per row (:167)."""
        self._fitted('scale', 'X_scl')
        return self._lazy('X_scl', lambda: self._scl_f[self._feature_rows()][:, None])

    @property
    def Ur(self):
        """(n_local, r) POD basis rows held by this rank."""
        return self._lazy('Ur', lambda: np.ascontiguousarray(self._engine().to_host(self._fitted('Ur', 'Ur'), result=True)))

    @Ur.setter
    def Ur(self, value):
        # subclasses written against the reference assign the result of decomposition() (gpr.py:386):
        # that array already has its device twin; anything else is uploaded
        self._flush_deferred()
        self._d.pop('nrm0', None)                             # row norms of the basis fit() stored, not of this one
        last = self.__dict__.get('_last_decomp')
        if last is not None and value is last[0]:
            self._d['Ur'] = last[1]
        else:
            self._d['Ur'] = self._engine().to_device(np.asarray(value, dtype=np.float64))
        self._host['Ur'] = value

    @property
    def X0(self):
        """(n_local, m) centred/scaled matrix (:169, :492); built on first access only."""
        def make():
            eng = self._engine()
            Xd, rowmean = self._Xd(), self._fitted('rowmean', 'X0')
            n, m = Xd.shape
            block = max(1, self._X0_BLOCK_BYTES // (8 * m))
            if n <= block or not hasattr(eng, '_to_host_staged'):
                return eng.to_host(eng.scale_rows(Xd, self._row0, self.n_points, self.n_features, rowmean, self._d['inv_scale']))
            # a scaled copy of a big X does not fit next to it in HBM (184 + 184 GB at BASELINE config 3): row blocks, each
            # scaled into a scratch block and streamed to its place in the host array
            out = np.empty((n, m), dtype=np.float64)
            for i0 in range(0, n, block):
                i1 = min(n, i0 + block)
                t = eng.scale_rows(Xd[i0:i1], self._row0 + i0, self.n_points, self.n_features, rowmean[i0:i1],
                                   self._d['inv_scale'])
                eng._to_host_staged(t, out=out[i0:i1])
            return out
        return self._lazy('X0', make)

    _X0_BLOCK_BYTES = 1 << 30      # X0 is materialised through device blocks of this size above it

    @X0.setter
    def X0(self, value):
        self._host['X0'] = value                              # gpr.py:379 stores what scale_data returned

    # ------------------------------------------------------------------ a2 scale_data
    _DEVICE_SCALINGS = ('std', 'none', 'pareto', 'vast', 'range', 'level', 'max', 'variance', 'median', 'poisson',
                        'l2-norm')

    def _check_scaling(self, scale_type, axis_cnt):
        known = ['std', 'none', 'pareto', 'vast', 'range', 'level', 'max', 'variance', 'median', 'poisson',
                 'vast_2', 'vast_3', 'vast_4', 'l2-norm']
        if scale_type not in known:
            raise NotImplementedError('The scaling method selected has not been implemented yet')   # :164
        if scale_type not in self._DEVICE_SCALINGS:
            # the reference's own 'vast_2/3/4' branches assign scipy's per-COLUMN kurtosis (an m-vector, :148 / :152 / :156)
            # to a slice of n_points rows: NumPy refuses that assignment with a ValueError unless m == n_points (or
            # m == 1, which broadcasts) -- the same exception here, before any device work
            m = self.X.shape[1]
            if m not in (1, self.n_points):
                raise ValueError(f'could not broadcast input array from shape ({m},) into shape ({self.n_points},)')
            raise NotImplementedError(f"scale_type={scale_type!r} has no device implementation (per-column kurtosis "
                                      'assigned to rows: only defined in the reference when n_points == m); no CPU fallback.')
        # axis_cnt is handed to np.average(x, axis=axis_cnt) on an (n_points, m) block whose result is assigned to n_points rows
        # (:112): 1 and -1 are the row means, None the block mean; 0 / -2 give m column means, which NumPy refuses to put into
        # n_points rows (ValueError) unless the two counts coincide; any other integer is not an axis of a 2-D block (AxisError)
        if axis_cnt is None or axis_cnt == 1 and not isinstance(axis_cnt, bool):
            return axis_cnt
        if isinstance(axis_cnt, (int, np.integer)) and not isinstance(axis_cnt, bool):
            if axis_cnt == -1:
                return 1
            if axis_cnt in (0, -2):
                m = self.X.shape[1]
                if m not in (1, self.n_points):
                    raise ValueError(f'could not broadcast input array from shape ({m},) into shape ({self.n_points},)')
            else:
                raise np.exceptions.AxisError(int(axis_cnt), 2, 'axis')      # (np.average names its argument: 'axis: axis 2 is out of ...')
        raise NotImplementedError(f'axis_cnt={axis_cnt!r}: row centring (1, -1) and scalar centring (None) have a '
                                  'device implementation; no CPU fallback for the rest.')

    def _feature_scale(self, scale_type, cnt, mu, var, m):
        """Per-feature scaling factor (:114-161) from the merged block statistics: cnt rows, block mean mu,
        population variance var of the raw block; 'range' / 'max' add one min/max pass over X."""
        with np.errstate(invalid='ignore', divide='ignore'):
            std = np.sqrt(var)
            if scale_type == 'std':
                return std
            if scale_type == 'none':
                return np.ones_like(std)
            if scale_type == 'pareto':
                return np.sqrt(std)
            if scale_type == 'vast':
                return var / mu
            if scale_type == 'level':
                return mu.copy()
            if scale_type == 'variance':
                return var.copy()
            if scale_type == 'poisson':
                return np.sqrt(mu)
            if scale_type == 'l2-norm':
                return np.sqrt(cnt * m * (var + mu * mu))
            if scale_type == 'median':
                return self._feature_median()
            eng = self._engine()
            mm = self._all_gather(eng.feature_minmax(self._Xd(), self._row0, self.n_points, self.n_features))
            mm = eng.to_host(mm)                              # (world, F, 2)
            fmin, fmax = mm[:, :, 0].min(axis=0), mm[:, :, 1].max(axis=0)
            return fmax - fmin if scale_type == 'range' else fmax

    _SELECT_DIGITS = (13, 13, 13, 13, 12)

    def _feature_median(self):
        """np.median of every raw feature block (:140-141) by radix selection on the device: five histogram passes
        over X (csrc/select.hip), the cumulative walk between passes on the host; histograms are all-reduced so
        every rank follows the same path.  Exact: returns the mean of the two middle order statistics."""
        eng = self._engine()
        Xd = self._Xd()
        F, m = self.n_features, Xd.shape[1]
        N = self.n_points * m                                  # values per feature block (global)
        want = np.array([(N - 1) // 2, N // 2], dtype=np.int64)
        rank = np.tile(want, (F, 1))                           # remaining rank inside the current prefix
        prefix = np.zeros((F, 2), dtype=np.uint64)
        shift = 64
        i64 = eng.torch.int64
        for bits in self._SELECT_DIGITS:
            shift -= bits
            two = bool(np.any(prefix[:, 0] != prefix[:, 1]))
            hist = eng.feature_digit_hist(Xd, self._row0, self.n_points, F, eng.to_device(prefix.view(np.int64), dtype=i64),
                                          shift, bits, two)
            hist = eng.to_host(self._all_reduce(hist))          # (F, 2, 1 << bits)
            cum = np.cumsum(hist, axis=2)
            for f in range(F):
                for t in range(2):
                    b = int(np.searchsorted(cum[f, t], rank[f, t], side='right'))
                    if b >= cum.shape[2]:
                        raise RuntimeError('median selection lost its target (NaN in X?)')
                    rank[f, t] -= cum[f, t, b - 1] if b > 0 else 0
                    prefix[f, t] |= np.uint64(b) << np.uint64(shift)
        sign = np.uint64(1) << np.uint64(63)
        bits_ = np.where(prefix & sign, prefix ^ sign, ~prefix)   # inverse of the order-preserving key
        vals = bits_.view(np.float64)
        return 0.5 * (vals[:, 0] + vals[:, 1])

    def _stats_pass(self, scale_type='std', axis_cnt=1):
        """Fused K1+K3a pass, cross-rank merge, per-feature scale. Leaves rowmean/scale on the device."""
        eng = self._engine()
        Xd = self._Xd()
        F = self.n_features
        tr_ = self._trace = _Trace(eng)
        fused = scale_type if (axis_cnt == 1 and scale_type in getattr(eng, 'SCALE_CODES', ())) else None
        rowmean, gram, fs_d = self._gram_collective(Xd, combine=fused)
        self._merge_stats(gram, fs_d, rowmean, scale_type, axis_cnt)

    def _gram_collective(self, Xd, combine=None):
        """The fused stats + Gram pass over the local rows and the ONE collective of fit(): an all-reduce (sum) of a
        buffer [F m m Gram doubles | world x F x 3 statistics | world first rows], every rank writing its (count, mean, M2)
        triples and its first global row into its own slots and zeros elsewhere, so the sum hands every rank all ranks'
        statistics and row blocks in rank order (adding zeros is exact) -- north_star: a single RCCL all-reduce for the
        Gram matrix.
        Returns rowmean (n_local,), gram (F, m, m) summed over ranks, fstats_all (world, F, 3)."""
        eng = self._engine()
        F, m = self.n_features, Xd.shape[1]
        fill = self._filler_wanted(Xd)
        e0 = eng.timing_event() if fill else None
        if not self._dist():
            rowmean, fstats, gram = eng.stats_gram(Xd, self._row0, self.n_points, F, center=True)
            if fill:
                self._gram_events = (e0, eng.timing_event())
            self._trace.mark('stats_gram')
            return rowmean, gram, fstats[None]
        world, rank = self._world(), self._shard.rank
        self.__dict__.pop('_combined', None)
        nc = self._native_comm()
        if nc is not None and combine is not None and m <= 512 and hasattr(eng, 'fit_gram_pass'):
            # the whole pass -- Gram kernel, finalize, all-reduce, statistics merge + scaled sum -- as ONE call of the library
            # (spr_fit_gram_pass): nothing of the host between the kernels and the collective; combine = the scale_type
            close = self._comm_bracket('allreduce')           # (brackets the whole enqueue here)
            import time
            self.last_comm_ = ('spr_fit_gram_pass (Gram + all_reduce + combine)', (F, m, m), time.time())
            rowmean, buf, packed_d, scale_d, inv_d = eng.fit_gram_pass(Xd, self._row0, self.n_points, F, combine, nc, world)
            close()
            if fill:
                self._gram_events = (e0, eng.timing_event())
            self._trace.mark('stats_gram')
            self._combined = (packed_d, scale_d, inv_d)
            fstats_all = buf[F * m * m:F * m * m + world * F * 3].view(world, F, 3)
            if '_layout' not in self.__dict__:
                self._layout_src = (fstats_all[:, :, 0], buf[F * m * m + world * F * 3:])
            return rowmean, buf[:F * m * m].view(F, m, m), fstats_all
        buf = eng.zeros((F * m * m + world * F * 3 + world,))
        # this rank's first row (exact below 2^53), see _shard_layout -- from a resident scalar: assigning a Python float is a
        # synchronous pageable H2D copy, which queues on an SDMA engine BEHIND the pushes of a p2p field exchange in flight
        # and held the next Gram pass back by their whole duration (profiles/r05_c4share8_p2p_loopback_timeline.txt)
        r0 = self.__dict__.get('_row0_d')                       # (a device tensor: never pickled, see __getstate__)
        if r0 is None:
            r0 = self._row0_d = eng.to_device(np.array([float(self._row0)]))
        buf[F * m * m + world * F * 3 + rank:F * m * m + world * F * 3 + rank + 1].copy_(r0)
        g_view = buf[:F * m * m].view(F, m, m)
        f_view = buf[F * m * m + rank * F * 3:F * m * m + (rank + 1) * F * 3].view(F, 3)
        if getattr(eng, 'supports_gram_out', False):          # the finalize kernel writes straight into the collective buffer
            rowmean, fstats, gram = eng.stats_gram(Xd, self._row0, self.n_points, F, center=True, gram_out=g_view,
                                                   fstats_out=f_view)
        else:                                                 # an engine without the output arguments (the NumPy test double)
            rowmean, fstats, gram = eng.stats_gram(Xd, self._row0, self.n_points, F, center=True)
            g_view.copy_(gram)
            f_view.copy_(fstats)
        if fill:
            self._gram_events = (e0, eng.timing_event())
        self._trace.mark('stats_gram')
        close = self._comm_bracket('allreduce')
        self._all_reduce(buf)
        close()
        fstats_all = buf[F * m * m:F * m * m + world * F * 3].view(world, F, 3)
        if '_layout' not in self.__dict__:
            # rows per (rank, feature) and first rows: the ranks' blocks, carried by the one all-reduce for free
            self._layout_src = (fstats_all[:, :, 0], buf[F * m * m + world * F * 3:])
        return rowmean, buf[:F * m * m].view(F, m, m), fstats_all

    def _filler_wanted(self, Xd):
        import os
        eng = self._engine()
        tr = self.__dict__.get('_trace')
        env = os.environ.get('SPR_GAP_FILLER')
        on = (self.gap_filler or env == '1') and env != '0'
        return bool(on and hasattr(eng, 'gram_filler') and self._GAP_FILL_MIN_M <= Xd.shape[1]
                    and not (tr is not None and tr.on))        # SPR_TRACE synchronises at every mark: its gaps are not fit()'s

    def _queue_gap_filler(self, Xd):
        """Called between the enqueue of the Gram download and the host's wait for it (see gap_filler)."""
        eng = self._engine()
        hist = self.__dict__.get('_gap_hist')
        ev = self.__dict__.pop('_gram_events', None)
        rate = self.__dict__.get('_gram_rows_per_ms')
        self._gap_fill_rows = 0
        self._gram_events_pending = ev
        # A field all-gather left in flight by the previous reconstruct(wait=False) cannot run NEXT TO the Gram or projection
        # workgroups (RCCL's kernel wants 261-280 VGPRs per wave and 19.7 KB of LDS; two Gram waves hold 448 of a SIMD's 512
        # registers, two projection waves all of them): its window is this very gap, so it stays empty
        pf = self.__dict__.get('_pending_field')
        if pf is not None and pf.pending and pf.needs_cus:     # (the p2p exchange runs on the SDMA engines: nothing to leave free)
            return
        if hist and rate and min(hist) >= self._GAP_FILL_MIN_MS:
            # a wide X (m > 256) is filled with the 256-column kernel on its first slice, whose rows cost (256 / m)^2 of what the
            # rows of the whole wide pass -- the rate measured -- cost
            m = Xd.shape[1]
            speed = (m / 256.0) ** 2 if m > 256 else 1.0
            rows = int(self._GAP_FILL_FRACTION * min(hist) * rate * speed) // 4096 * 4096
            rows = min(rows, Xd.shape[0])
            if rows >= 65536:
                eng.gram_filler(Xd, rows, self._row0, self.n_points, self.n_features)
                self._gap_fill_rows = rows

    def _close_gap(self):
        """The projection is about to be launched: record the host gap this fit() had, and the rate of its Gram pass."""
        t0 = self.__dict__.pop('_gap_t0', None)
        if t0 is None:
            return
        import time
        from collections import deque
        eng = self._engine()
        gap_ms = 1e3 * (time.perf_counter() - t0)
        hist = self.__dict__.setdefault('_gap_hist', deque(maxlen=8))
        hist.append(gap_ms)
        ev = self.__dict__.pop('_gram_events_pending', None)
        if ev is not None:
            ms = eng.elapsed_ms(*ev)                     # both events lie in front of the download the host has waited for
            if ms > 0:
                self._gram_rows_per_ms = self._Xd().shape[0] / ms

    def _merge_stats(self, gram, fs_d, rowmean, scale_type, axis_cnt):
        """Per-feature Gram blocks (all-reduced) + per-rank feature statistics -> X_scl per feature and the Gram matrix
        G of X0 = (X - X_cnt)/X_scl (self._G, host)."""
        eng = self._engine()
        Xd = self._Xd()
        m = Xd.shape[1]
        F = self.n_features
        tr_ = self._trace
        if axis_cnt == 1 and scale_type in getattr(eng, 'SCALE_CODES', ()) and hasattr(eng, 'gram_combine'):
            # statistics merge, feature scales and G = sum_f G_f / scl_f^2 on the device (csrc/combine.hip): one
            # download of m^2 + 5F doubles, the scales never leave HBM
            comb = self.__dict__.pop('_combined', None)       # spr_fit_gram_pass has merged already (native communicator)
            packed_d, scale_d, inv_d = comb if comb is not None else eng.gram_combine(gram, fs_d, scale_type)
            fill = self._filler_wanted(Xd) and getattr(self, '_fit_fills_gap', False)

            def gap_hook():
                # the download is enqueued, the host is about to block on it: what goes into the gap behind it -- a
                # reconstruct the caller deferred (useful work), else the filler (discarded work, opt-in)
                if self._flush_deferred():
                    self._gap_fill_rows = 0                    # (the gap holds the deferred launch, not a filler)
                elif fill:
                    self._queue_gap_filler(Xd)
            packed = eng.to_host(packed_d, then=gap_hook)
            if fill:
                import time
                self._gap_t0 = time.perf_counter()
            tr_.mark('collect')
            feat = packed[m * m:].reshape(F, 5)
            self._G = packed[:m * m].reshape(m, m)
            self._scl_f = feat[:, 3].copy()
            self._var_f = self._scl_f ** 2
            self._mu_f = feat[:, 1].copy()
            self._set_precenter_ratio(feat[:, 1], feat[:, 2], feat[:, 4], axis_cnt)
            self._d['rowmean'] = rowmean
            self._d['scale'] = scale_d
            self._d['inv_scale'] = inv_d
            for k in ('X_cnt', 'X_scl', 'X0'):
                self._host.pop(k, None)
            tr_.mark('merge')
            return
        # one download for both: every host round trip is a sync point the GPU idles at
        packed = eng.to_host(eng.torch.cat([gram.reshape(-1), fs_d.reshape(-1)]), then=self._flush_deferred)
        G_f = packed[:gram.numel()].reshape(F, m, m)
        fs = packed[gram.numel():].reshape(fs_d.shape[0], F, 3)
        tr_.mark('collect')
        cnt = np.zeros(F); mu = np.zeros(F); m2 = np.zeros(F)
        for w in range(fs.shape[0]):                         # Chan merge in rank order
            nb, mb, sb = fs[w, :, 0], fs[w, :, 1], fs[w, :, 2]
            tot = cnt + nb
            with np.errstate(invalid='ignore', divide='ignore'):
                d = mb - mu
                frac = np.where(tot > 0, nb / np.where(tot > 0, tot, 1), 0.0)
                m2 = m2 + sb + d * d * cnt * frac
                mu = mu + d * frac
            cnt = tot
        tr = np.trace(G_f, axis1=1, axis2=2)
        # a feature WITHOUT rows -- only possible in a partial row group (RowShard(partial=True): one rank's block of a
        # larger job run alone) -- takes no part: scale 1, and its Gram block is zero anyway
        present = cnt > 0
        with np.errstate(invalid='ignore', divide='ignore'):
            var_f = np.where(present, (tr + m * m2) / np.where(present, cnt * m, 1.0), 0.0)   # population variance (:115)
        scl = self._feature_scale(scale_type, cnt, mu, var_f, m)
        self._scl_f = np.where(present, scl, 1.0)
        self._var_f = self._scl_f ** 2                        # what the Gram blocks are divided by
        self._mu_f = mu.copy()
        with np.errstate(invalid='ignore', divide='ignore'):
            fluct = np.where(present, tr / np.where(present, cnt * m, 1.0), 0.0)
        self._set_precenter_ratio(mu, var_f, fluct, axis_cnt)
        if axis_cnt is None:
            # scalar centre per feature (:112 with axis=None): turn the row-centred Gram blocks into those
            # of (X - mu_f) with the two column-sum vectors, and make X_cnt the per-feature constant
            cs = eng.to_host(self._all_reduce(eng.colsums(Xd, self._row0, self.n_points, F, rowmean)))
            v = cs[:, 1, :] - mu[:, None] * cs[:, 0, :]       # sum_i (mean_i - mu_f) c_i
            G_f = G_f + v[:, :, None] + v[:, None, :] + m2[:, None, None]
            rowmean = eng.fill_feature(Xd.shape[0], self._row0, self.n_points, eng.to_device(mu))
        with np.errstate(invalid='ignore', divide='ignore'):
            self._G = np.sum(G_f / self._var_f[:, None, None], axis=0)   # Gram matrix of X0 = (X - X_cnt)/X_scl
        self._d['rowmean'] = rowmean
        self._d['scale'] = eng.to_device(self._scl_f)
        with np.errstate(divide='ignore'):
            self._d['inv_scale'] = eng.to_device(1.0 / self._scl_f)
        for k in ('X_cnt', 'X_scl', 'X0'):
            self._host.pop(k, None)
        tr_.mark('merge')

    # The projection kernels remove the row mean in their epilogue, x.W - mean (1^T W): the products carry the mean
    # through the MFMA and lose log10(|mean| / |x - mean|) digits, which W = V/S amplifies by sigma_1/sigma_i for the
    # small modes.  Above this product the mean is subtracted from the operand before the multiplication instead
    # (spr_project_stream_* centre mode 2) -- the reference's own order of operations (:169).
    _PRECENTER_ABOVE = 1e6

    def _set_precenter_ratio(self, mu, var, fluct, axis_cnt):
        """Per feature: how large the centre the projection's epilogue has to cancel is, relative to what is left after the
        cancellation.  Row centring (axis_cnt = 1): the rows' own means -- the block mean plus four standard deviations
        of the row means, sqrt(var - fluct), as a stand-in for max_i |mean_i| -- over the rms of the row-centred values,
        sqrt(fluct) = sqrt(trace(G_f) / (count m)): the block std would not do, it contains the spread of the means.
        Scalar centring (axis_cnt = None): X_cnt is the block mean itself and what remains is the whole block's spread."""
        mu, var, fluct = (np.asarray(a, dtype=np.float64) for a in (mu, var, fluct))
        with np.errstate(invalid='ignore', divide='ignore'):
            if axis_cnt is None:
                self._pc_ratio = np.abs(mu) / np.sqrt(var)
            else:
                self._pc_ratio = (np.abs(mu) + 4.0 * np.sqrt(np.maximum(var - fluct, 0.0))) / np.sqrt(fluct)

    def _needs_precenter(self, kappa):
        ratio = getattr(self, '_pc_ratio', None)
        if ratio is None or not np.isfinite(kappa):
            return False
        ratio = ratio[np.isfinite(ratio)]
        return bool(ratio.size and ratio.max() * kappa > self._PRECENTER_ABOVE)

    def scale_data(self, scale_type='std', axis_cnt=1):
        """Reference :83-171.  Sets X_cnt / X_scl and returns the scaled matrix X0."""
        self._flush_deferred()
        axis_cnt = self._check_scaling(scale_type, axis_cnt)
        self._stats_pass(scale_type, axis_cnt)
        return self.X0

    def scale_limits(self, limits):
        """Reference :173-210 (the method GPR users call on the base class, tests/test_gpr_data.py:95): per-feature
        limits -> per-row scaled limits [(limit_f - X_cnt)/X_scl], with the reference's +-1000 block clamps.
        Host arithmetic on the (n_local,) centring vector; X_scl is one scalar per feature."""
        X_cnt = self.X_cnt[:, 0]
        feat = self._feature_rows()
        out = []
        for limit in limits:
            limit = np.asarray(limit, dtype=np.float64)
            limit0 = (limit[feat] - X_cnt) / self._scl_f[feat]
            for f in range(self.n_features):                  # :200-203, per feature block (local rows of it)
                sel = feat == f
                if sel.any():
                    if limit0[sel].min() < -1000:
                        limit0[sel] = -1000
                    elif limit0[sel].max() > 1000:
                        limit0[sel] = 1000
            out.append(limit0)
        return out

    def _csr_device(self, C, known=None):
        """(indptr, indices, vals) device tensors of a dense / scipy.sparse matrix with n columns."""
        import scipy.sparse as sp
        eng = self._engine()
        if known is not None:
            indptr, indices, vals = known
        elif isinstance(C, OneHotRows):
            if C.ndim != 2:
                raise ValueError('a measurement matrix has two dimensions')
            indptr, indices, vals = np.arange(C.shape[0] + 1), C.rows, np.ones(C.shape[0])
        else:
            Cs = C.tocsr() if sp.issparse(C) else sp.csr_matrix(np.asarray(C, dtype=np.float64))
            Cs.sort_indices()
            indptr, indices, vals = Cs.indptr, Cs.indices, Cs.data
        t = eng.torch
        return (eng.to_device(indptr, dtype=t.int64), eng.to_device(indices, dtype=t.int64), eng.to_device(vals))

    def _sampled(self, sampling, matmul=False):
        """S.Ur, S.X_cnt, S.X_scl for a sampling matrix S (s, n) -- reference :233, :366."""
        if sampling.shape[1] != self._n_global:
            # NumPy's own text, as the reference's first product with S raises it: sampling.dot(X_cnt) in reconstruct (:366-368),
            # sampling @ X_cnt in unscale_data (:233)
            n, c = self._n_global, sampling.shape[1]
            if matmul:
                raise ValueError('matmul: Input operand 1 has a mismatch in its core dimension 0, with gufunc signature '
                                 f'(n?,k),(k,m?)->(n?,m?) (size {n} is different from {c})')
            raise ValueError(f'shapes ({sampling.shape[0]},{c}) and ({n},1) not aligned: {c} (dim 1) != {n} (dim 0)')
        eng = self._engine()
        ip, ix, v = self._csr_device(sampling)
        Th, cnt, scl = eng.measure_csr(ip, ix, v, self._fitted('Ur', 'Ur'), self._row0, self._fitted('rowmean', 'X_cnt'),
                                       scale=self._d['scale'], n_points=self.n_points)
        return self._all_reduce(Th), self._all_reduce(cnt), self._all_reduce(scl)

    # ------------------------------------------------------------------ a11 unscale_data
    def unscale_data(self, x0, sampling=None):
        """Reference :212-240: x = X_scl * x0 + X_cnt for an (n_local,) vector, or with ``sampling`` (s, n)
        x = (S X_scl) * x0 + S X_cnt for an (s,) vector."""
        self._flush_deferred()
        if type(x0) is not np.ndarray:
            raise NotImplementedError('unscale_data of a cvxpy expression is outside the device path.')
        eng = self._engine()
        if sampling is not None:
            _, cnt, scl = self._sampled(sampling, matmul=True)
            ones = eng.to_device(np.ones(1))
            t = eng.unscale(eng.to_device(x0), 0, x0.shape[0], 1, cnt, ones, rowscale=scl)
        else:
            t = eng.unscale(eng.to_device(x0), self._row0, self.n_points, self.n_features,
                            self._fitted('rowmean', 'X_scl'), self._d['scale'])        # (:235 reads X_scl first)
        return eng.to_host(t)

    # ------------------------------------------------------------------ a4 reduction
    def _select_rank(self, exp_variance, n_cols, select_modes, n_modes):
        """Integer logic of ROM.reduction (:314-333), same exceptions."""
        if select_modes == 'variance':
            if not 0 <= n_modes <= 100:
                raise ValueError('The parameter n_modes is outside the[0-100] range.')
            if n_modes == 100:
                r = n_cols
            else:
                r = 1
                while exp_variance[r - 1] < n_modes:
                    r += 1
        elif select_modes == 'number':
            if not type(n_modes) is int:
                raise TypeError('The parameter n_modes is not an integer.')
            if not 1 <= n_modes <= n_cols:
                raise ValueError('The parameter n_modes is outside the [1-m] range.')
            r = n_modes
        else:
            raise ValueError('The select_mode value is wrong.')
        return r

    def reduction(self, U, A, exp_variance, select_modes, n_modes):
        """Reference :281-340 (host arrays in, views out)."""
        r = self._select_rank(exp_variance, A.shape[1], select_modes, n_modes)
        self.r = r
        return U[:, :r], A[:, :r]

    # ------------------------------------------------------------------ a3 decomposition
    @staticmethod
    def _expvar(lam):
        lam_pos = np.maximum(lam, 0.0)
        return 100 * np.cumsum(lam_pos) / np.sum(lam_pos)      # :274-275

    def _eig_local(self, G, rank_of):
        """This rank's eigen-solve of the (m, m) Gram matrix -> (lam descending (m,), V with r or m columns, unsigned).
        The top-r route (dsytrd + dsterf + r inverse iterations + dormqr) for m >= 96 when the caller only needs r <= m/2
        vectors and sigma_1/sigma_r is within the plain Gram route's range (the refinement pass needs all of V)."""
        m = G.shape[0]
        if (rank_of is not None and _eigen._EIGH_TOP_NATIVE_MIN_M <= m < _eigen._EIGH_TOP_MIN_M
                and getattr(rank_of, 'known_r', None) is not None):
            # the number of modes is given (select_modes='number'): the whole route in one library call
            r = rank_of.known_r
            if 2 * r <= m:
                got = _eigen._eig_top_native(G, r)
                if got is not None and np.isfinite(got[0][0]) and not np.sqrt(max(got[0][r - 1], 0.0)) * _GRAM_KAPPA_REFINE < np.sqrt(max(got[0][0], 0.0)):
                    return got
        if rank_of is not None and m >= _eigen._EIGH_TOP_MIN_M:
            lam_a, fac = _eigen._eigh_tridiagonal(G)
            lam = lam_a[::-1].copy()
            S = np.sqrt(np.maximum(lam, 0.0))
            r = rank_of(self._expvar(lam))
            if 2 * r <= m and np.isfinite(S[0]) and not S[r - 1] * _GRAM_KAPPA_REFINE < S[0]:
                V = _eigen._eigvecs_top(fac, lam_a, r)
                if V is not None:
                    return lam, V
        lam, V = _eigen._eigh_small(G)
        return lam[::-1].copy(), np.ascontiguousarray(V[:, ::-1])

    def _spectrum(self, G, rank_of=None):
        """Eigen-decomposition of the (m,m) Gram matrix -> S (desc), V, explained variance (:272-275).
        ``rank_of(exp_variance) -> r``: the caller only needs the r leading vectors -- V then has r columns whenever
        the top-r route applies (_eig_local) and all m otherwise.
        Sharded: every rank solves for itself (identical bits in, identical bits out on one node: no exchange), checked
        ONCE per object by a digest all-gather at the first fit(); if the ranks differ -- or with
        RowShard(broadcast_basis=True) -- only rank 0 solves and its factors are broadcast in one fixed-size message
        [route, columns, lam (m), V (m x m, padded)], so that no rank-local decision (the route, r, a failed inverse
        iteration) can change the size or the branch of a collective."""
        if not np.all(np.isfinite(G)):
            # a constant feature (X_scl = 0 -> X0 = nan/inf, :169) or a NaN/Inf in X: np.linalg.svd(X0) (:272) raises
            raise np.linalg.LinAlgError('SVD did not converge')
        m = G.shape[0]
        if self._broadcasts_basis():
            lam, V = self._broadcast_factors(G, rank_of, None)
        else:
            lam, V = self._eig_local(G, rank_of)
            if self._dist() and not self.__dict__.get('_basis_checked', False):
                self._basis_checked = True
                if not self._factors_agree(lam, V):
                    import sys
                    self._basis_diverged = True
                    if self._shard.rank == 0:
                        print('[openmeasure_amd] the ranks\' host eigen-solves of the same Gram matrix differ (hosts or LAPACK '
                              'builds differ): rank 0\'s factors are broadcast from now on', file=sys.stderr)
                    lam, V = self._broadcast_factors(G, rank_of, (lam, V))
        self.basis_broadcast_ = bool(self._broadcasts_basis())
        return np.sqrt(np.maximum(lam, 0.0)), _eigen._sign_fix(V), self._expvar(lam)

    def _refine_spectrum(self, S, V, r, Xd, row0, n_points, n_features, inv_scale_d, rowmean_d, center, rank_of=None):
        """Conditioning safeguard of the Gram route (hard part 1).

        The eigenvectors of G = X0^T X0 carry an error eps * (sigma_1/sigma_i)^2, which LAPACK's SVD of X0 itself
        (reference :272) does not have.  One more pass over X removes it: with the first-stage factors V^, S^,
        Y = X0 V^ diag(1/S^) has columns of nearly unit norm that are nearly orthogonal, so its Gram matrix
        H = Y^T Y (same MFMA kernels: projection of a row block into an f64 scratch block, Gram of the block) is
        formed with errors relative to 1, not to sigma_1^2 -- the small modes are now resolved to eps * sigma_1/sigma_i
        like in the reference.  From H = Z L Z^T:  X0 = (Y Z L^-1/2) (L^1/2 Z^T diag(S^) V^T) = Q M with Q
        orthonormal, and the SVD of the m x m matrix M gives the singular values and right singular vectors of X0.
        Repeated (at most _GRAM_REFINE_MAX_PASSES passes) until the retained block of H is well conditioned;
        raises LinAlgError if it never is -- never returns silently degraded sensors.
        ``rank_of``: exp_variance -> number of modes the caller will keep (so that the last pass may compute only that many
        right singular vectors; columns r .. m of the V returned are zero then).
        Returns (S, V, exp_variance, passes)."""
        eng = self._engine()
        n_loc, m = Xd.shape
        eps = np.finfo(float).eps
        # row blocks of the f64 scratch matrix Y: up to 8 GiB (a quarter of what is free), equal in size, so that a pass over 9M
        # rows is 3 projection + Gram launch pairs instead of 9 (each pair has its ramp and tail; one 18 GB block measured no
        # better than three of 6 GB, and its allocation now and then cost 5 ms)
        budget = 1 << 31
        try:
            free = eng.torch.cuda.mem_get_info(eng.device)[0] if hasattr(eng, 'device') and eng.device.type == 'cuda' else 0
            budget = int(min(8 << 30, max(1 << 31, free // 4)))
        except (RuntimeError, AttributeError):
            pass
        block = int(max(1 << 16, min(n_loc, budget // (8 * (m + (m & 1))))))
        n_blocks = -(-n_loc // block)
        block = -(-n_loc // n_blocks)                         # equal blocks: no short last pair of launches (ramp and tail for little work)
        block = min(n_loc, -(-block // 16) * 16)
        Y = eng.empty((min(block, n_loc), m + (m & 1)))
        passes = 0
        import time
        prof = self.refine_profile_ = dict(device_ms=0.0, host_ms=0.0)
        while True:
            passes += 1
            t_a = time.perf_counter()
            floor = S[0] * np.sqrt(m * eps)
            d = np.maximum(S, floor if floor > 0 else 1.0)
            W2 = eng.to_device(V / d)
            pre = bool(center and self._needs_precenter(S[0] / d[r - 1]))
            H_d = None
            for i0 in range(0, n_loc, block):
                rows = min(block, n_loc - i0)
                eng.project_f64(Xd, i0, rows, row0, n_points, n_features, inv_scale_d, W2, rowmean_d, Y, center=center,
                                precenter=pre)
                Yb = Y[:rows, :m] if Y.shape[1] != m else Y[:rows]
                _, _, g = eng.stats_gram(Yb, 0, rows, 1, center=False)
                H_d = g[0].clone() if H_d is None else H_d.add_(g[0])
            H = eng.to_host(self._all_reduce(H_d))
            t_b = time.perf_counter()
            H = 0.5 * (H + H.T)
            # X0 = Q M with Q orthonormal: from the Cholesky factor H = R^T R (Q = Y R^-1, M = R diag(d) V^T: 0.3 ms at
            # m = 256 against 2.7 ms for the eigen-decomposition H = Z L Z^T, M = L^1/2 Z^T diag(d) V^T, which remains the
            # route when H is not numerically positive definite -- a first stage too far off, the null mode of a full-rank fit)
            with _eigen._one_blas_thread():                           # m x m: a many-core BLAS pool only gets in the way
                from scipy.linalg import lapack
                Rc, info = lapack.dpotrf(H, lower=0, clean=1)
                if info == 0 and np.all(np.isfinite(Rc)) and Rc.diagonal().min() > 1e-7 * Rc.diagonal().max():
                    t_c = time.perf_counter()
                    B = Rc * d[None, :]
                else:
                    lamH, Z = _eigen._eigh_small(H)
                    t_c = time.perf_counter()
                    B = (np.sqrt(np.maximum(lamH, 0.0))[:, None] * Z.T) * d[None, :]
                # M = B V^T, V orthogonal: the singular values of M are B's, its right singular vectors V times B's
                t_d = time.perf_counter()
                # of this SVD only the singular values and the r retained right vectors are used once the pass has converged
                # (the usual case): the one-call route computes exactly those (0.65 of dgesdd's time at m = 256, r = 64);
                # rows r .. m of Vt stay zero then.  A pass that has NOT converged needs all of V for the next one (below).
                # (not when the ranks take rank 0's factors: the route must not depend on a rank-local outcome there)
                part = None if self._broadcasts_basis() else _eigen._svd_top_native(B, r)
                if part is not None:
                    S_new = part[0]
                    Vt = np.zeros((m, m))
                    Vt[:r] = (V @ part[1]).T                    # m x m x r instead of forming M (m x m x m)
                else:
                    M = B @ V.T
                    _, S_new, Vt = np.linalg.svd(M)
            t_e = time.perf_counter()
            prof['eigh_ms'] = prof.get('eigh_ms', 0.0) + 1e3 * (t_c - t_b)
            prof['M_ms'] = prof.get('M_ms', 0.0) + 1e3 * (t_d - t_c)
            prof['svd_ms'] = prof.get('svd_ms', 0.0) + 1e3 * (t_e - t_d)
            S_new, Vt = self._same_on_all_ranks(S_new, Vt)
            prof['device_ms'] += 1e3 * (t_b - t_a)
            # is the pass converged?  The retained columns of Y must have come out nearly orthonormal.  Modes below
            # 1e-12 sigma_1 (the null mode that row-centring creates when all m modes are kept) are rounding noise
            # in the reference as well and are left out of the verdict.
            keep = np.flatnonzero(S_new[:r] > 1e-12 * S_new[0])
            dn = np.sqrt(np.maximum(np.diag(H), np.finfo(float).tiny))
            Hk = (H[np.ix_(keep, keep)] / dn[keep, None]) / dn[None, keep]     # unit diagonal
            rho = float(np.max(np.sum(np.abs(Hk), axis=1) - np.abs(np.diag(Hk)))) if len(keep) else 0.0
            if rho < 0.5 and np.all(np.isfinite(Hk)):
                # Gershgorin: the eigenvalues of the retained block lie in [1 - rho, 1 + rho] -- condition below 3 without solving
                # for them (the usual case: off-diagonals of 1e-3)
                cond_r = (1.0 + rho) / (1.0 - rho)
            else:
                with _eigen._one_blas_thread():
                    ev = np.linalg.eigvalsh(Hk)
                cond_r = ev[-1] / max(ev[0], np.finfo(float).tiny)
            converged = cond_r < 4.0                           # |off-diagonal| of the retained block well below 1
            if part is not None and (not converged or (rank_of is not None and rank_of(self._expvar(S_new * S_new)) > r)):
                # all of V is needed after all: another pass follows, or the refined spectrum asks for more modes than were kept
                with _eigen._one_blas_thread():
                    _, S_new, Vt = np.linalg.svd(B @ V.T)
            S, V = S_new, _eigen._sign_fix(Vt.T.copy())
            prof['host_ms'] += 1e3 * (time.perf_counter() - t_b)
            if converged:
                break
            if passes >= _GRAM_REFINE_MAX_PASSES:
                raise np.linalg.LinAlgError(
                    f'fit: sigma_1/sigma_r = {S[0] / max(S[r - 1], np.finfo(float).tiny):.3g} is beyond what the Gram '
                    f'route resolves even after {passes} refinement passes (retained block of the second-stage Gram '
                    f'matrix still has condition {cond_r:.3g}); keep fewer modes or rescale the data.')
        lam = S * S
        exp_variance = 100 * np.cumsum(lam) / np.sum(lam)
        return S, V, exp_variance, passes

    def _basis_from_gram(self, G, select_modes, n_modes, center, inv_scale_d):
        eng = self._engine()
        Xd = self._Xd()
        m = Xd.shape[1]
        import time
        t_eig = time.perf_counter()
        def rank_of(ev):
            return self._select_rank(ev, m, select_modes, n_modes)
        # select_modes='number': r is known before the eigenvalues are (the one-call top-r route, _eigen._eig_top_native)
        rank_of.known_r = n_modes if (select_modes == 'number' and type(n_modes) is int and 1 <= n_modes <= m) else None
        S, V, exp_variance = self._spectrum(G, rank_of)
        self.eig_ms_ = 1e3 * (time.perf_counter() - t_eig)      # host wall time of the m x m eigen-solve (bench.py: per rank)
        self._trace.mark('eigh')
        r = self._select_rank(exp_variance, m, select_modes, n_modes)
        self.gram_refine_passes_ = 0
        if S[r - 1] * _GRAM_KAPPA_REFINE < S[0]:
            import time
            t_ref = time.perf_counter()
            S, V, exp_variance, self.gram_refine_passes_ = self._refine_spectrum(
                S, V, r, Xd, self._row0, self.n_points, self.n_features, inv_scale_d, self._d.get('rowmean'), center,
                rank_of=rank_of)
            r = self._select_rank(exp_variance, m, select_modes, n_modes)
            self.refine_ms_ = 1e3 * (time.perf_counter() - t_ref)      # host wall time of the refinement (it synchronises)
            self._trace.mark('refine')
        # modes below sqrt(m eps) sigma_1 carry no information on the Gram route; keep the
        # projection finite for them (their reference counterparts are LAPACK rounding noise)
        floor = S[0] * np.sqrt(m * np.finfo(float).eps) if not self.gram_refine_passes_ else S[0] * m * np.finfo(float).eps
        S_safe = np.maximum(S[:r], floor if floor > 0 else 1.0)
        W = V[:, :r] / S_safe
        self._trace.mark('W')
        # (W's previous reader is the projection of the previous fit(), in front of the Gram download the host has waited for)
        W_d = eng.upload_reuse(('W', id(self)), W) if hasattr(eng, 'upload_reuse') else eng.to_device(W)
        self._trace.mark('upload')
        self.precentered_ = bool(center and self._needs_precenter(S[0] / S_safe[-1]))
        nrm0 = self._norms_buffer(Xd, r, self.precentered_ and center)
        kw = {} if nrm0 is None else {'norms': nrm0}
        self._close_gap()
        self._flush_deferred()                                # (normally done in the gap already: see _merge_stats)
        Ur_d = eng.project(Xd, self._row0, self.n_points, self.n_features, inv_scale_d, W_d,
                           center=center, out=self._d.pop('Ur', None), rowmean=self._d.get('rowmean'),
                           basis_dtype=self._basis_dtype(), precenter=self.precentered_, **kw)
        if nrm0 is not None:
            self._d['nrm0'] = nrm0
        self._trace.mark('project')
        Ar = V[:, :r] * S[:r]                                # A = (diag(S) Vt).T  (:273)
        return Ur_d, Ar, exp_variance[:r], S, r, V[:, :r]

    def decomposition(self, X0, select_modes='variance', n_modes=99):
        """Reference :242-279 on a caller-supplied scaled matrix X0 (host ndarray, local rows).
        Returns (Ur, Ar, exp_variance[:r]) as host arrays.  When X0 is the very array scale_data() returned
        (the pattern of GPR.fit, gpr.py:379-381) the Gram blocks of that pass and the resident X are used instead
        of uploading X0 and reading it twice."""
        self._flush_deferred()
        eng = self._engine()
        if X0 is self._host.get('X0') and '_G' in self.__dict__ and 'rowmean' in self._d:
            G = self._G
            self._trace = _Trace(eng)
            self._d.pop('Ur', None)
            self._d.pop('nrm0', None)
            Ur_d, Ar, expv, _, r, _ = self._basis_from_gram(G, select_modes, n_modes, True, self._d['inv_scale'])
        else:
            X0d = eng.to_device(X0)
            _, _, gram = eng.stats_gram(X0d, 0, X0d.shape[0], 1, center=False)
            G = eng.to_host(self._all_reduce(gram))[0]
            ones = eng.to_device(np.ones(1))
            m = X0d.shape[1]
            S, V, exp_variance = self._spectrum(G, lambda ev: self._select_rank(ev, m, select_modes, n_modes))
            r = self._select_rank(exp_variance, m, select_modes, n_modes)
            floor = S[0] * np.sqrt(m * np.finfo(float).eps)
            if S[r - 1] * _GRAM_KAPPA_REFINE < S[0]:
                S, V, exp_variance, _ = self._refine_spectrum(S, V, r, X0d, 0, X0d.shape[0], 1, ones, None, False,
                                                              rank_of=lambda ev: self._select_rank(ev, m, select_modes, n_modes))
                r = self._select_rank(exp_variance, m, select_modes, n_modes)
                floor = S[0] * m * np.finfo(float).eps
            W = V[:, :r] / np.maximum(S[:r], floor if floor > 0 else 1.0)
            Ur_d = eng.project(X0d, 0, X0d.shape[0], 1, ones, eng.to_device(W), center=False)
            Ar, expv = V[:, :r] * S[:r], exp_variance[:r]
        self.r = r
        Ur = np.ascontiguousarray(eng.to_host(Ur_d))
        self._last_decomp = (Ur, Ur_d)
        return Ur, Ar, expv

    # ------------------------------------------------------------------ a5 fit
    def fit(self, scale_type='std', axis_cnt=1, select_modes='variance', n_modes=99, basis=None):
        """Reference :463-511."""
        axis_cnt = self._check_scaling(scale_type, axis_cnt)
        if basis is None and select_modes not in ('variance', 'number'):
            raise ValueError('The select_mode value is wrong.')
        eng = self._engine()
        self.scale_type = scale_type
        for k in ROM._LAZY + ('C', 'Theta', '_pending'):
            self.__dict__.pop(k, None)
        for k in ('cnt', 'Theta', 'nrm0'):                    # a new basis invalidates the trained measurement state
            self._d.pop(k, None)
        took = self._device_fit(scale_type, axis_cnt, select_modes, n_modes, basis)
        if took is True:
            return
        if took != 'merged':                                  # 'merged': the device route left its statistics behind
            self._fit_fills_gap = basis is None               # the host eigen-solve follows: see gap_filler
            try:
                self._stats_pass(scale_type, axis_cnt)
            finally:
                self._fit_fills_gap = False
        self._host.clear()
        if basis is None:
            Ur_d, Ar, expv, S, r, V_r = self._basis_from_gram(self._G, select_modes, n_modes, True, self._d['inv_scale'])
            self.exp_variance_ = expv
            self.S_ = S
        else:
            self._flush_deferred()
            Ur_d = eng.to_device(basis[0])
            Ar = np.asarray(basis[1])
            V_r = None
        self._d['Ur'] = Ur_d
        if '_layout_src' in self.__dict__ and '_layout' not in self.__dict__:
            # first sharded fit(): the table of row blocks rode on the all-reduce -- check now that the blocks tile the global
            # rows (ValueError on every rank), not at the first gather: placement, train and predict use global indices too
            self._shard_layout(Ur_d.shape[0])
        self.Ar = Ar
        self.r = Ar.shape[1]
        Sigma_r = np.linalg.norm(Ar, axis=0)                  # :504-508
        self.Sigma_r = Sigma_r
        if V_r is not None:
            self.Vr = V_r * (np.linalg.norm(V_r, axis=0) ** -1)   # = Ar / Sigma_r, also when a sigma underflowed to 0
        else:
            self.Vr = Ar / Sigma_r
        self._trace.report()

    def _device_fit(self, scale_type, axis_cnt, select_modes, n_modes, basis):
        """fit() without any host synchronisation: statistics merge, feature scales, eigen-decomposition
        (Jacobi, csrc/spectrum.hip) and W = V_r S_r^-1 all stay on the device.  Taken when the spectrum fits one
        workgroup and beats the host round trip (m <= 24), the number of modes is given, and the scaling derives
        from block mean/variance;
        Ar, Sigma_r, Vr, exp_variance_, X_scl are copied to the host on first access."""
        eng = self._engine()
        if basis is not None or select_modes != 'number' or axis_cnt != 1 or not hasattr(eng, 'spectrum'):
            return False
        if scale_type not in eng.SCALE_CODES:
            return False
        Xd = self._Xd()
        m = Xd.shape[1]
        if m > min(eng.spectrum_max_m, _DEVICE_SPECTRUM_MAX_M):
            return False
        r = self._select_rank(None, m, 'number', n_modes)      # same TypeError / ValueError as the reference
        F = self.n_features
        tr_ = self._trace = _Trace(eng)
        rowmean, gram, fs_all = self._gram_collective(Xd)
        sp = eng.spectrum(gram, fs_all, scale_type, r)
        tr_.mark('stats_gram+spectrum')
        self._host.clear()
        self._d['rowmean'] = rowmean
        self._d['scale'] = sp['scale']
        self._d['inv_scale'] = sp['inv_scale']
        nrm0 = self._norms_buffer(Xd, r)
        kw = {} if nrm0 is None else {'norms': nrm0}
        self._flush_deferred()
        self._d['Ur'] = eng.project(Xd, self._row0, self.n_points, F, sp['inv_scale'], sp['W'], center=True,
                                    out=self._d.pop('Ur', None), rowmean=rowmean, basis_dtype=self._basis_dtype(), **kw)
        if nrm0 is not None:
            self._d['nrm0'] = nrm0
        tr_.mark('project')
        # the only download of this path: the two singular values that decide whether the Gram route was good enough,
        # the Jacobi verdict (sweeps, off^2, diag^2) and the feature statistics -- fetched after the projection has been
        # enqueued, so the device never idles on the good path
        chk = eng.to_host(eng.torch.cat([sp['S'][:1], sp['S'][r - 1:r], sp['info'], sp['feat'].reshape(-1)]))
        feat = chk[5:].reshape(F, 5)
        self._scl_f, self._mu_f = feat[:, 3].copy(), feat[:, 1].copy()
        self._set_precenter_ratio(feat[:, 1], feat[:, 2], feat[:, 4], axis_cnt)
        good = bool(np.all(np.isfinite(chk)))
        if good:
            converged = chk[2] < eng.spectrum_max_sweeps or chk[3] <= 1e-24 * chk[4]
            kappa = chk[0] / chk[1] if chk[1] > 0 else np.inf
            good = converged and not kappa > _GRAM_KAPPA_REFINE and not self._needs_precenter(kappa)
        if not good:
            # out of the plain Gram route's range (or non-finite data): the host route decides -- refinement pass,
            # pre-centred projection or LinAlgError -- from the Gram blocks this pass already has (no second read of X)
            self._merge_stats(gram, fs_all, rowmean, scale_type, axis_cnt)
            self._device_fit_fallback_ = True
            return 'merged'
        self.r = r
        self.gram_refine_passes_ = 0
        self.precentered_ = False
        sp['r'] = r
        self._pending = sp
        tr_.report()
        return True

    # ------------------------------------------------------------------ a10 reconstruct
    def reconstruct(self, Ar, sampling=None, to_host=True, wait=True):
        """Reference :342-375.  Returns X_rec of shape (n, n_p) (all ranks' rows, gathered).

        ``to_host=False`` returns the device tensor of shape (n_p, n) instead (same values,
        column-major) and skips the PCIe copy.  With ``wait=False`` as well, a PendingField comes back
        right after the all-gather of the field has been ENQUEUED, so the gather (720 MB per rank at config 4)
        runs on the communication stream under whatever the caller launches next -- e.g. the MFMA-bound Gram
        pass of the next fit(); call ``.wait()`` before reading it.  (With ``defer_reconstruct``, the default, that form only
        RECORDS its launch: see the attribute.)"""
        eng = self._engine()
        self._flush_deferred()                                # an earlier deferred launch keeps its place in the order
        Ar = np.asarray(Ar, dtype=np.float64) if not hasattr(Ar, 'is_cuda') else Ar
        if Ar.ndim < 2:
            Ar = Ar[None, :]
        if Ar.shape[0] == 0:                                  # no measurement vectors: (n, 0), nothing to launch
            rows = self._n_global if sampling is None else sampling.shape[0]
            return np.zeros((rows, 0)) if to_host else eng.empty((0, rows))
        A_d = Ar if hasattr(Ar, 'is_cuda') else eng.to_device(Ar)
        if sampling is not None:                              # :365-368 -- (S Ur) Ar^T, un-scaled with S X_scl, S X_cnt
            Th, cnt, scl = self._sampled(sampling)
            ones = eng.to_device(np.ones(1))
            Thp = Th if Th.shape[1] % 2 == 0 else eng.torch.nn.functional.pad(Th, (0, 1))[:, :Th.shape[1]]
            out = eng.reconstruct(Thp, 0, Th.shape[0], 1, cnt, ones, A_d, rowscale=scl)
            return out if not to_host else eng.to_host(out, result=True).T
        Ur_d = self._fitted('Ur', 'Ur')
        self._fitted('rowmean', 'X_cnt')
        # the basis, centre and scale this object holds NOW: what the launch works on, whenever it is enqueued
        state = (Ur_d, self._d['rowmean'], self._d['scale'])
        if self._defers() and not to_host and not wait:
            # defer_reconstruct: record the launch instead (the captured tensors stay alive with it)
            n_loc, n_p = Ur_d.shape[0], A_d.shape[0]
            if hasattr(Ar, 'is_cuda'):
                A_d = A_d.clone()                             # the caller's own tensor: what it holds NOW (n_p x r doubles), whatever
                                                              # the caller writes into it before the launch
            total = int(self._shard_layout(n_loc)[:, 1].sum()) if self._dist() else n_loc
            pf = self._deferred = PendingField(None, launch=lambda: self._reconstruct_now(A_d, state, False, False),
                                               shape=(n_p, total), needs_cus=False)
            return pf
        return self._reconstruct_now(A_d, state, to_host, wait)

    def _defers(self):
        import os
        env = os.environ.get('SPR_DEFER_RECONSTRUCT')
        return (self.defer_reconstruct or env == '1') and env != '0'

    def _reconstruct_now(self, A_d, state, to_host, wait, path=None):
        """Enqueue the reconstruct kernel (and, sharded, the exchange of the field) on ``state`` = (Ur, rowmean, scale) device
        tensors.  ``path``: 'p2p' / 'rccl' for this call only (the first-exchange trial), None: the object's choice."""
        eng = self._engine()
        Ur_d, rowmean_d, scale_d = state
        n_loc = Ur_d.shape[0]
        n_p = A_d.shape[0]
        world = self._world()
        if not self._dist():
            if to_host and hasattr(eng, 'reconstruct_to_host'):
                # the reference's contract (:371-375): a host ndarray.  Big fields go out in row chunks whose copies run
                # under the next chunk's kernel, into page-locked memory (engine.reconstruct_to_host)
                host = eng.reconstruct_to_host(Ur_d, self._row0, self.n_points, self.n_features, rowmean_d, scale_d, A_d)
                if host is not None:
                    return host.T
            out = eng.reconstruct(Ur_d, self._row0, self.n_points, self.n_features, rowmean_d, scale_d, A_d)
        else:
            import torch.distributed as dist
            lay = self._shard_layout(n_loc)
            if (path or self._gather_select(n_p, lay)) == 'p2p':
                return self._reconstruct_p2p(A_d, state, lay, to_host, wait)
            if np.any(lay[:, 1] != n_loc):
                return self._gather_unequal(A_d, state, lay, to_host, wait)
            loc = eng.reconstruct(Ur_d, self._row0, self.n_points, self.n_features, rowmean_d, scale_d, A_d)
            # ONE all-gather for all n_p columns: rank q's (n_p, n_loc) block lands at stage[q]; for one column that is
            # the field itself, for several the columns are put side by side afterwards -- on the way to the host when
            # the caller wants a host array (block copies, no pass over the field on the device), by
            # spr_field_unstage_f64 when the field stays in HBM
            stage = eng.empty((world, n_p, n_loc))
            close = self._comm_bracket('gather')              # issue -> join, when the join happens inside this call
            import time
            self.last_comm_ = ('field all_gather (rccl)', (world, n_p, n_loc), time.time())
            nc = self._native_comm()
            if nc is not None:
                # the library's own communicator: the all-gather on a side stream behind the reconstruct kernel, joined by an event
                t = eng.torch
                side = self.__dict__.get('_comm_stream')
                if side is None:
                    side = self._comm_stream = t.cuda.Stream(eng.device)
                ev0 = t.cuda.Event()
                ev0.record(t.cuda.current_stream(eng.device))
                side.wait_event(ev0)
                with t.cuda.stream(side):
                    eng.comm_allgather(nc, loc, stage)
                    ev1 = t.cuda.Event()
                    ev1.record(side)
                loc.record_stream(side)
                stage.record_stream(side)

                class _Joined:
                    @staticmethod
                    def wait():
                        t.cuda.current_stream(eng.device).wait_event(ev1)
                work = _Joined()
            else:
                work = dist.all_gather_into_tensor(stage.view(-1), loc.contiguous().view(-1), group=self._shard.group,
                                                   async_op=True)
            if n_p == 1:
                out = stage.view(1, world * n_loc)
                if not to_host and not wait:
                    pf = PendingField(out, [work], keep=(loc, stage),
                                      on_wait=lambda: self._comm_bracket('gather_exposed'))
                    self._pending_field = pf                  # fit() leaves its host gap free while this is in flight
                    return pf
                work.wait()
                close()
            else:
                work.wait()
                close()
                if to_host and hasattr(eng, 'stage_to_host'):
                    host = eng.stage_to_host(stage)
                    if host is not None:
                        return host.T
                out = eng.field_unstage(stage)
                if not to_host and not wait:
                    return PendingField(out)
        if not to_host:
            return out if wait else PendingField(out)
        return eng.to_host(out, result=True).T                             # (n, n_p), Fortran-ordered view

class SPR(GemPlacement, ROM):
    """Sparse Placement for Reconstruction (reference: SPR, sparse_sensing.py:513-901)."""

    placement_norms = None      # "auto": see ROM.placement_norms

    #: optimal_placement('qr') on one rank: between two passes over the whole basis, refresh only the rows whose norm can
    #: still win a step (pivot_loop / _pivot_loop_pooled).  Same sensors; False: one full sweep per batch of steps.
    placement_pools = True

    def __init__(self, X, n_features, xyz, shard=None, engine=None):
        super().__init__(X, n_features, xyz, shard=shard, engine=engine)

    def _check_rank_cap(self, what):
        """The placement and solve kernels are built for bases of up to SPR_MAX_R_WIDE columns (the reference takes any
        r <= m, :336, :739): say so BEFORE any device work, naming the cap."""
        if self.r > SPR_MAX_R_WIDE:
            raise ValueError(f'{what}: r = {self.r} retained modes exceed the {SPR_MAX_R_WIDE} the placement and solve '
                             'kernels are built for (fit and reconstruct take any r); keep fewer modes.')

    # ------------------------------------------------------------------ a6 optimal_placement
    def optimal_placement(self, calc_type='qr', n_sensors=10, mask=None, d_min=0., verbose=False):
        """Reference :700-756.  Returns the one-hot measurement matrix C of shape (s, n): the reference's dense ndarray
        while it is small (64 MiB), a OneHotRows above that (46 GB dense at BASELINE config 3) -- same behaviour for
        C[i, :], C @ x, np.argmax(C, axis=1), train(C).  The ordered global sensor rows are also in ``self.sensors_``."""
        self._flush_deferred()
        if calc_type == 'gem':
            return self._placement_gem(n_sensors, mask, d_min, verbose)
        if calc_type != 'qr':
            raise NotImplementedError('The sensor selection method has not been implemented yet')
        eng = self._engine()
        n = self._n_global
        Ur_d = self._fitted('Ur', 'Ur')
        self._check_rank_cap('optimal_placement')
        if mask is not None:
            mask = np.asarray(mask)
            _check_mask(mask, Ur_d.shape[0])
            eng.mask_rows(Ur_d, eng.to_device(mask.astype(np.uint8), dtype=eng.torch.uint8))   # :737-738
            self._host.pop('Ur', None)
            self._d.pop('nrm0', None)                         # rows were zeroed: the norms fit() left no longer hold
        s = self.r
        nrm0 = self._d.get('nrm0')                            # left by fit() under placement_norms, for this very basis
        self.placement_from_norms_ = nrm0 is not None and nrm0.shape[0] == Ur_d.shape[0]
        if self.placement_from_norms_:
            st = eng.qr_begin(Ur_d, self._row0, s, norms=nrm0)
        else:
            st = eng.qr_begin(Ur_d, self._row0, s)
        stats = {}
        sweeps = pivot_loop(eng, st, s, self._all_gather if self._dist() else None, pools=self.placement_pools,
                            stats=stats)
        self.pivot_sweeps_ = sweeps - int(self.placement_from_norms_)   # passes over the basis (the start read none)
        self.pivot_pool_sweeps_ = stats.get('pool_sweeps', 0)           # refreshes that visited the pool only
        self.pivot_log_ = stats.get('log', [])                          # batches and refreshes of the pooled driver, in order
        piv = eng.to_host(st['piv']).astype(np.int64)
        self.sensors_ = piv
        self.pivot_gap_ = eng.to_host(st['gap'])
        C = self._one_hot(piv, n)
        self._placed = (C, piv)
        return C

    @staticmethod
    def _one_hot(piv, n):
        """(s, n) one-hot matrix of the picks (:741-743): the reference's dense ndarray while it is small, a OneHotRows
        (same behaviour for the documented uses, s integers of storage) above _DENSE_C_LIMIT bytes."""
        C = OneHotRows(piv, n)
        return C.toarray() if len(piv) * n * 8 <= _DENSE_C_LIMIT else C

    # ------------------------------------------------------------------ a7 train
    def train(self, C, is_Theta=False, limits=None, method='OLS', solver='CLARABEL', cond=False, verbose=False):
        """Reference :758-820."""
        self._flush_deferred()
        n = self._n_global
        if (C.shape[1] != n) and not is_Theta:
            raise ValueError('The number of columns of C does not match the number'
                             ' of rows of X.')
        if method == 'COLS':
            raise NotImplementedError("method='COLS' needs a conic solver; it has no device implementation "
                                      '(no CPU fallback).')
        eng = self._engine()
        if not is_Theta:
            placed = getattr(self, '_placed', None)
            known = None
            if placed is not None and placed[0] is C:
                piv = placed[1]
                known = (np.arange(len(piv) + 1), piv, np.ones(len(piv)))
            ip, ix, v = self._csr_device(C, known)
            Theta_d, cnt_d = eng.measure_csr(ip, ix, v, self._fitted('Ur', 'Ur'), self._row0,
                                             self._fitted('rowmean', 'X_cnt'))
            Theta_d = self._all_reduce(Theta_d)
            cnt_d = self._all_reduce(cnt_d)
            self.C = C
            self._d['cnt'] = cnt_d
            Theta = eng.to_host(Theta_d)
        else:
            Theta = np.asarray(C, dtype=np.float64)
            Theta_d = None
        if Theta.shape[1] != self.r:
            raise ValueError('The number of columns of Theta does not match the number'
                             ' of columns of Ur.')
        self._d['Theta'] = Theta_d if Theta_d is not None else eng.to_device(Theta)
        self.Theta = Theta
        self.limits = limits
        self.method = method
        self.solver = solver
        self.verbose = verbose
        if cond == True:                                      # noqa: E712  (:813-820; s x r, host-sized)
            if Theta.shape[0] == Theta.shape[1]:
                S_theta = np.linalg.svd(Theta, compute_uv=False)
            else:
                S_theta = np.linalg.svd(np.linalg.pinv(Theta), compute_uv=False)
            self.k = S_theta[0] / S_theta[-1]

    # ------------------------------------------------------------------ a8 / a9
    _PINV_RCOND = 1e-15    # np.linalg.pinv's default in the reference's NumPy (:873, :877)

    def _solve(self, ys):
        """scale_vector + (weighted) least squares for a list of measurement vectors, on the device.
        Full-column-rank, well-conditioned systems take the MFMA normal-equations kernel; everything else the
        reference's pinv accepts -- fewer sensors than modes (every GEM placement), rank-deficient W Theta,
        cond^2 beyond what the refined normal equations resolve -- takes the QR + one-sided-Jacobi SVD kernel
        with the same rcond cut, which returns the minimum-norm solution exactly as np.linalg.pinv does."""
        eng = self._engine()
        if 'cnt' not in self._d:
            raise AttributeError("'SPR' object has no attribute 'C'")      # reference fails at self.C (:573)
        Theta_d, cnt_d = self._d['Theta'], self._d['cnt']
        self._check_rank_cap('predict')
        if cnt_d.shape[0] != Theta_d.shape[0]:
            # train(Theta, is_Theta=True) after train(C) with another sensor count: C.dot(X_cnt) (:573) no longer
            # matches the rows of y -- the reference fails with a broadcast error at :578
            raise ValueError(f'operands could not be broadcast together: y has {Theta_d.shape[0]} rows, '
                             f'C has {cnt_d.shape[0]}')
        Yh = np.stack([np.asarray(y, dtype=np.float64) for y in ys])
        fid = Yh[:, :, 2].astype('int')                                   # :576 indexes X_scl with it
        n_loc_rows = self._n_global
        bad = (fid * self.n_points >= n_loc_rows) | (fid * self.n_points < -n_loc_rows)
        if bad.any():
            k = int(fid.ravel()[np.argmax(bad.ravel())]) * self.n_points
            raise IndexError(f'index {k} is out of bounds for axis 0 with size {n_loc_rows}')
        if (fid < 0).any():                                               # numpy wraps negative indices (:576)
            Yh = Yh.copy()
            Yh[:, :, 2] = np.where(fid < 0, fid + self.n_features, fid)
        Y = eng.to_device(Yh)
        s, r = Theta_d.shape
        if s >= r and r <= getattr(eng, 'ols_max_r', r):      # the normal-equations kernel keeps its factor in LDS
            Ar_d, As_d, y0_d, info_d = eng.solve_ols(Theta_d, cnt_d, self._d['scale'], Y)
            many = getattr(eng, 'to_host_views', None)        # the four outputs in ONE download (they share a buffer)
            got = many(info_d, Ar_d, As_d, y0_d) if many is not None else None
            info = got[0] if got is not None else eng.to_host(info_d)
            if np.any(info[:, 0] == 2):
                # an uncertainty that is zero (or NaN) for SOME sensors of a vector: W = diag(1/0) (:872) and
                # np.linalg.pinv(W @ Theta) (:873) raises -- flagged by the kernel, no second solve
                raise np.linalg.LinAlgError('SVD did not converge')
            # the kernel's refinement step (corrected semi-normal equations) is as accurate as a QR solve while
            # cond(W Theta)^2 eps < 1; info[:, 1] estimates cond^2 from the Cholesky pivots
            if not (np.any(info[:, 0] != 0) or np.any(info[:, 1] > 1e13) or not np.all(np.isfinite(info[:, 1]))):
                self.solve_path_ = 'cholesky'
                if got is not None:
                    return got[1], got[2], got[3]
                return eng.to_host(Ar_d), eng.to_host(As_d), eng.to_host(y0_d)
        Ar_d, As_d, y0_d, info_d = eng.solve_pinv(Theta_d, cnt_d, self._d['scale'], Y, rcond=self._PINV_RCOND)
        many = getattr(eng, 'to_host_views', None)
        got = many(info_d, Ar_d, As_d, y0_d) if many is not None else None
        info = got[0] if got is not None else eng.to_host(info_d)
        if np.any(info[:, 0] < 0):
            raise np.linalg.LinAlgError('SVD did not converge')           # what np.linalg.pinv raises
        self.solve_path_ = 'pinv'
        self.solve_rank_ = info[:, 1].astype(int)
        if got is not None:
            return got[1], got[2], got[3]
        return eng.to_host(Ar_d), eng.to_host(As_d), eng.to_host(y0_d)

    def scale_vector(self, y):
        """Reference :553-584.  Returns y0 (s,2); sets cnt_vector / scl_vector."""
        _, _, y0 = self._solve([y])
        self.cnt_vector = self._engine().to_host(self._d['cnt'])
        self.scl_vector = self._scl_f[np.asarray(y)[:, 2].astype('int')]
        return y0[0]

    def predict(self, y):
        """Reference :822-901 (OLS branch).  Returns (Ar, Ar_sigma), each (n_p, r)."""
        if isinstance(y, np.ndarray):
            y = [y]
        for i in range(len(y)):
            if self.Theta.shape[0] != y[i].shape[0]:
                raise ValueError('The number of rows of Theta does not match the number'
                                 ' of rows of y.')
            if y[i].shape[1] != 3:
                raise ValueError('The y array has the wrong number of columns. y has'
                                 ' to have dimensions (s,3).')
        if self.method != 'OLS':
            raise NotImplementedError('The prediction method selected has not been '
                                      'implemented yet')
        if len(y) == 0:                                       # :863-864 allocate (0, r) and the loop never runs
            return np.zeros((0, self.r)), np.zeros((0, self.r))
        Ar, Ar_sigma, y0 = self._solve(y)
        self.cnt_vector = self._engine().to_host(self._d['cnt'])
        self.scl_vector = self._scl_f[np.asarray(y[-1])[:, 2].astype('int')]
        return Ar, Ar_sigma
