"""Host eigen / SVD routes of fit() (reference: the V factor of np.linalg.svd, sparse_sensing.py:272, of which :336 keeps r
columns): LAPACK on the m x m Gram matrix between the two passes over X -- dsyevd, the top-r route (dsytrd + dsterf + batched
inverse iteration + dormqr, in Python or as one native call of libspr_hip.so), and the top-r SVD that ends a refinement pass.
Module-level names are looked up at call time (tests patch them here)."""
from __future__ import annotations

import numpy as np

_BLAS_CTL = None


def _one_blas_thread():
    """Context manager capping the BLAS pool at one thread: the m x m problems between the two passes over X are far too
    small for a many-core pool (128 threads made the 256 x 256 eigen-solve 10x slower on the GPU host).  The
    threadpoolctl controller is built once -- discovering the loaded BLAS libraries costs more than the solve."""
    global _BLAS_CTL
    if _BLAS_CTL is None:
        try:
            from threadpoolctl import ThreadpoolController
            _BLAS_CTL = ThreadpoolController()
        except ImportError:                               # pragma: no cover
            _BLAS_CTL = False
    if not _BLAS_CTL:
        import contextlib
        return contextlib.nullcontext()
    return _BLAS_CTL.limit(limits=1, user_api='blas')       # tools/eigh_probe.py: 1 thread is fastest for m <= 512


def _eigh_small(G):
    """All eigenpairs of the symmetric (m, m) matrix G (ascending): LAPACK dsyevd called directly on one BLAS thread --
    2.65 ms at m = 256 on the GPU host against 3.0 ms through np.linalg.eigh (and 4 ms / 95 ms at m = 256 / 512 with the
    host's default 128-thread pool)."""
    from scipy.linalg import lapack
    with _one_blas_thread():
        w, v, info = lapack.dsyevd(np.asarray(G, dtype=np.float64).T, compute_v=1, lower=1)   # G.T: Fortran view, no copy
    if info != 0:
        raise np.linalg.LinAlgError('Eigenvalues did not converge')
    return w, v


_EIGH_TOP_MIN_M = 96           # below this dsyevd is a fraction of a millisecond: nothing to gain (m = 64, r = 32: dsyevd 156-173 us,
                               # the top-r route 140 us in isolation but 207 us inside fit() on a second host -- round 5)
_EIGH_TOP_NATIVE_MIN_M = 32    # ... between the two, ONE native call (spr_host_eig_top): m = 64, r = 32 150 us against 173 us for dsyevd,
                               # m = 41, r = 14 63 against 86; from m = 128 on its plain loops lose to the BLAS calls of the Python route
_LWORK = {}


def _eigh_tridiagonal(G):
    """First half of the top-r route: G = Q T Q^T (dsytrd) and ALL eigenvalues of T (dsterf, ascending).
    -> (lam, factorisation) ; raises LinAlgError like _eigh_small."""
    from scipy.linalg import lapack
    m = G.shape[0]
    with _one_blas_thread():
        lw = _LWORK.get(('trd', m))
        if lw is None:
            lw = _LWORK[('trd', m)] = int(lapack.dsytrd_lwork(m, lower=1)[0])
        c, d, e, tau, info = lapack.dsytrd(np.asarray(G, dtype=np.float64).T, lower=1, lwork=lw)
        if info == 0:
            lam, info = lapack.dsterf(d, e)
    if info != 0:
        raise np.linalg.LinAlgError('Eigenvalues did not converge')
    return lam, (c, d, e, tau)


def _tridiag_vectors_batched(d, e, w):
    """Eigenvectors of the tridiagonal matrix (d, e) for the eigenvalues w, all inverse iterations side by side
    (spr_host_tridiag_vectors, csrc/host_eig.hip: the recurrences of dstein vectorised over the eigenvalue index -- 0.2 instead of
    0.77 ms for 64 of 256 on the GPU host).  It does not re-orthogonalise inside clusters: the vectors are accepted when they are
    orthonormal to 1e-8 as they come (separated eigenvalues) and then made so to rounding by one symmetric correction
    Z (I - (Z^T Z - I) / 2); otherwise None, and the caller takes dstein.  -> (m, r) Fortran-ordered array or None."""
    try:
        from . import _lib
        lib = _lib.load()
    except (RuntimeError, OSError, AttributeError):
        return None
    m, r = d.shape[0], w.shape[0]
    d, e, w = (np.ascontiguousarray(a, dtype=np.float64) for a in (d, e, w))
    Z = np.empty((m, r))
    if lib.spr_host_tridiag_vectors(d.ctypes.data, e.ctypes.data, m, w.ctypes.data, r, Z.ctypes.data, 4) != 0:
        return None
    E = Z.T @ Z
    E[np.diag_indices(r)] -= 1.0
    if not np.all(np.isfinite(E)) or np.abs(E).max() > 1e-8:
        return None
    return np.asfortranarray(Z - 0.5 * (Z @ E))


_LAPACK_PTRS = None


def _lapack_pointers():
    """Addresses of SciPy's LAPACK routines dsytrd / dsterf / dormtr (scipy.linalg.cython_lapack exports them as C function
    pointers in capsules); False when they cannot be had."""
    global _LAPACK_PTRS
    if _LAPACK_PTRS is None:
        try:
            import ctypes
            from scipy.linalg import cython_lapack
            get_name = ctypes.pythonapi.PyCapsule_GetName
            get_name.restype, get_name.argtypes = ctypes.c_char_p, [ctypes.py_object]
            get_ptr = ctypes.pythonapi.PyCapsule_GetPointer
            get_ptr.restype, get_ptr.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_char_p]
            out = []
            for name in ('dsytrd', 'dsterf', 'dormtr'):
                cap = cython_lapack.__pyx_capi__[name]
                ptr = get_ptr(cap, get_name(cap))
                if not ptr:
                    raise ValueError(name)
                out.append(ptr)
            _LAPACK_PTRS = tuple(out)
        except Exception:                                  # noqa: BLE001 -- any SciPy without these capsules: the Python route
            _LAPACK_PTRS = False
    return _LAPACK_PTRS


_LAPACK_SVD_PTRS = None


def _lapack_svd_pointers():
    """dgebrd / dbdsdc / dormbr of SciPy's LAPACK, as in _lapack_pointers; False when they cannot be had."""
    global _LAPACK_SVD_PTRS
    if _LAPACK_SVD_PTRS is None:
        try:
            import ctypes
            from scipy.linalg import cython_lapack
            get_name = ctypes.pythonapi.PyCapsule_GetName
            get_name.restype, get_name.argtypes = ctypes.c_char_p, [ctypes.py_object]
            get_ptr = ctypes.pythonapi.PyCapsule_GetPointer
            get_ptr.restype, get_ptr.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_char_p]
            out = []
            for name in ('dgebrd', 'dbdsdc', 'dormbr'):
                cap = cython_lapack.__pyx_capi__[name]
                ptr = get_ptr(cap, get_name(cap))
                if not ptr:
                    raise ValueError(name)
                out.append(ptr)
            _LAPACK_SVD_PTRS = tuple(out)
        except Exception:                                  # noqa: BLE001 -- any SciPy without these capsules: np.linalg.svd
            _LAPACK_SVD_PTRS = False
    return _LAPACK_SVD_PTRS


_SVD_TOP_MIN_M = 96      # below: the full dgesdd is a fraction of a millisecond


def _svd_top_native(M, r):
    """All singular values and the r leading RIGHT singular vectors of the square matrix M in one host call of the library
    (spr_host_svd_top: dgebrd, dbdsdc for the values, batched inverse iteration on the Golub-Kahan form, dormbr) -- what the
    refinement pass uses of np.linalg.svd(M), at 0.6 of its time for r = m / 4.  -> (S descending (m,), V (m, r)) or None (no
    library / no pointers / r too close to m / the vectors failed their checks: the caller takes np.linalg.svd)."""
    m = M.shape[0]
    if M.shape != (m, m) or m < _SVD_TOP_MIN_M or 2 * r > m:
        return None
    ptrs = _lapack_svd_pointers()
    if not ptrs:
        return None
    try:
        from . import _lib
        lib = _lib.load()
    except (RuntimeError, OSError, AttributeError):
        return None
    M = np.ascontiguousarray(M, dtype=np.float64)
    if not np.all(np.isfinite(M)):
        return None
    S, V = np.empty(m), np.empty((m, r))
    with _one_blas_thread():
        rc = lib.spr_host_svd_top(M.ctypes.data, m, r, S.ctypes.data, V.ctypes.data, *ptrs)
    if rc != 0 or not (np.all(np.isfinite(V)) and np.all(np.isfinite(S))):
        return None
    return S, V


def _eig_top_native(G, r):
    """The top-r route in ONE host call of the library (spr_host_eig_top: dsytrd, dsterf, the batched inverse iterations and
    dormtr back to back, LAPACK reached through SciPy's function pointers): at small m the route is mostly call overhead -- m = 64,
    r = 32: 173 us for dsyevd, 140-207 us for the same four steps glued in Python.  -> (lam descending (m,), V (m, r)) or None
    (no library / no pointers / the vectors failed their checks: the caller goes on with the Python route)."""
    ptrs = _lapack_pointers()
    if not ptrs:
        return None
    try:
        from . import _lib
        lib = _lib.load()
    except (RuntimeError, OSError, AttributeError):
        return None
    m = G.shape[0]
    G = np.ascontiguousarray(G, dtype=np.float64)
    lam, V = np.empty(m), np.empty((m, r))
    with _one_blas_thread():
        rc = lib.spr_host_eig_top(G.ctypes.data, m, r, lam.ctypes.data, V.ctypes.data, *ptrs)
    if rc == 1:
        raise np.linalg.LinAlgError('Eigenvalues did not converge')
    if rc != 0 or not np.all(np.isfinite(V)):
        return None
    return lam, V


def _eigvecs_top(fac, lam, r):
    """Second half: eigenvectors of the r LARGEST eigenvalues only -- inverse iteration on the tridiagonal matrix (all r at
    once in spr_host_tridiag_vectors; LAPACK's dstein, with its re-orthogonalisation inside clusters, when those fail their
    check) and back-transformation of the r vectors (dormqr on the reflectors dsytrd left below the sub-diagonal).  O(m^2 r)
    instead of dsyevd's O(m^3): dsytrd 0.61 + dsterf 0.36 + vectors and dormqr 0.57 = 1.6 ms against 2.65 ms at m = 256,
    r = 64 on the GPU host (tools/eigh_pieces_probe.py, profiles/r04_eigh_pieces_probe.txt).  -> V (m, r), columns in
    DESCENDING order of eigenvalue, or None when dstein reports a failure or the vectors are not orthonormal to 1e-12 (the
    caller then takes dsyevd)."""
    from scipy.linalg import lapack
    c, d, e, tau = fac
    m = d.shape[0]
    w = np.ascontiguousarray(lam[m - r:])
    iblock = np.ones(m, dtype=np.int32)
    isplit = np.zeros(m, dtype=np.int32)
    isplit[0] = m
    with _one_blas_thread():
        Z = _tridiag_vectors_batched(d, e, w)
        if Z is None:
            # LAPACK's dstein: one eigenvalue after the other, with re-orthogonalisation inside clusters (splitting its list
            # over a few threads gained nothing on the GPU host: 1.02 -> 1.09 ms, round 4)
            z, info = lapack.dstein(d, e, w, iblock, isplit)
            if info != 0:
                return None
            Z = np.asfortranarray(z[:, :r])
        cq = np.asfortranarray(c[1:, :m - 1])
        lw = _LWORK.get(('mqr', m, r))
        if lw is None:
            lw = _LWORK[('mqr', m, r)] = int(lapack.dormqr('L', 'N', cq, tau, np.asfortranarray(Z[1:]), lwork=-1)[1][0])
        out, _, info = lapack.dormqr('L', 'N', cq, tau, np.asfortranarray(Z[1:]), lwork=lw)
        if info != 0:
            return None
        V = np.ascontiguousarray(np.vstack([Z[:1], out])[:, ::-1])      # contiguous first: the check below then runs in BLAS
        if not np.all(np.isfinite(V)) or np.abs(V.T @ V - np.eye(r)).max() > 1e-12:
            return None
    return V


def _sign_fix(V):
    """Deterministic eigenvector signs: the entry of largest magnitude is positive."""
    idx = np.argmax(np.abs(V), axis=0)
    sgn = np.sign(V[idx, np.arange(V.shape[1])])
    sgn[sgn == 0] = 1.0
    return V * sgn
