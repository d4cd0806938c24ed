"""MI355X-native ``ROM`` / ``SPR`` with the class surface of
``openmeasure.sparse_sensing`` (reference: src/openmeasure/sparse_sensing.py).

Same constructor, method names, keyword defaults, return shapes and exception types as
the reference for the SPR hot path

    fit -> optimal_placement('qr') -> train -> predict('OLS') -> reconstruct

but the arithmetic runs in hand-written gfx950 kernels behind ``libspr_hip.so``
(include/spr_hip.h) instead of NumPy/LAPACK:

  reference call (file:line)                       here
  -----------------------------------------------  ------------------------------------------
  np.average / np.std / X0 = (X-cnt)/scl           one fused pass: row means, per-feature
    (:112, :115, :169)                               Welford stats, per-feature f64-MFMA Gram
  np.linalg.svd(X0) (:272)                         m x m eigen-problem of the Gram matrix (host,
                                                     tiny) + MFMA projection Ur = X0 V S^-1
  scipy.linalg.qr(Ur.T, pivoting=True) (:739)      greedy residual-norm pivoting on a candidate
                                                     set, a handful of MFMA sweeps over Ur
  SPR.gem (:586-698)                               the same pivoting on row-centred rows with the
                                                     search mask and d_min exclusion
  C.dot(Ur), C.dot(X_cnt) (:797, :573)             CSR row gather / SpMM
  np.linalg.pinv(W Theta) ... (:873-878)           MFMA normal equations + Cholesky per vector
  Ur @ Ar.T, unscale_data (:371-373, :235)         one streaming GEMV with fused un-scaling

What differs from the reference, by design:
  * X0 is never materialised by ``fit`` (``self.X0`` is built on first access);
  * singular vectors come from the Gram route, so columns of Ur/Ar/Vr may differ from
    LAPACK's by a sign, and a mode whose singular value is below ~1e-7 sigma_1 (the null
    mode that row-centring always creates when r = m) is numerically meaningless in both;
  * fitted arrays live in HBM; ``X_cnt``, ``X_scl``, ``Ur``, ... are copied to NumPy on
    first access;
  * a float32 X is stored as float32 in HBM; all arithmetic is float64 and the basis Ur is float64, as in the reference
    (X_cnt / X_scl are float64, :106-107, so X0 = (X - X_cnt)/X_scl and its SVD are float64 whatever the dtype of X);
    a float32 BASIS is an explicit storage option, DeviceMatrix(tensor, basis='f32') (BASELINE config 5);
  * 'gem' placement is the noise-free limit of the reference's rule (it adds unseeded noise, :667) for the first
    r-1 sensors and, beyond, a deterministic ridge stand-in for that noise (see _gem_ridge_phase);
  * fit() -> optimal_placement('gem') returns the GEM sensors OF THIS BASIS: the reference's rule takes the variance of
    a row over its r entries (:622, :638), which changes with the sign of a column of Ur, and LAPACK's signs are
    arbitrary -- the reference's own sensors coincide only after fit(basis=(Ur_ref, Ar_ref)) ('qr' placement, predict
    and reconstruct do not depend on the signs);
  * options that have no device implementation ('COLS', the kurtosis scalings 'vast_2..4', which are
    ill-defined in the reference itself) raise ``NotImplementedError`` -- they never fall back to a CPU path.

Row sharding: pass ``shard=RowShard(row0, n_global, group)`` and the local block of rows;
the Gram matrix is all-reduced, pivot candidates are all-gathered per step, Theta is
all-reduced (RCCL through torch.distributed) and every rank's block of the reconstructed field reaches every rank --
pushed by the SDMA engines into copies of the field the ranks of a node have mapped from each other (openmeasure_amd/p2p.py),
or all-gathered over RCCL (``RowShard(gather=...)``).
"""
from __future__ import annotations

# The implementation lives in four modules (round 6): rom.py (the class surface), _eigen.py (host eigen / SVD routes), _placement.py
# (pivot-loop drivers, GEM), _shard.py (row sharding, collectives, field exchange).  This module is the import point the
# reference's users know -- ``from openmeasure_amd.sparse_sensing import SPR`` -- and re-exports the public names.
from ._shard import PendingField, RowShard
from ._placement import pivot_loop
from .rom import ROM, SPR, DeviceMatrix, OneHotRows

__all__ = ['ROM', 'SPR', 'RowShard', 'DeviceMatrix', 'PendingField', 'OneHotRows', 'pivot_loop']
