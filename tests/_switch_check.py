"""Run by tests/test_gpu_parity.py::test_ab_switches_keep_parity in a process of its own (the library reads its A/B switches once):
the whole path on a few shapes against the oracle -- spectrum, ordered sensors, field -- under whatever SPR_* variables are set."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import spr_oracle as orc                                   # noqa: E402
from openmeasure_amd.engine import HipEngine                           # noqa: E402
from tests.parity import REL_FRO, rel_fro                              # noqa: E402
from tests.test_gpu_parity import _full_path, synth_host               # noqa: E402

eng = HipEngine('cuda:0')
for n_points, F, m, r in ((6000, 3, 64, 32), (5000, 2, 256, 64), (2000, 9, 41, 14), (1500, 2, 300, 24), (3000, 2, 128, 16)):
    rho = 10 ** (-3 / (r - 1))
    X = synth_host(n_points, F, m, min(m, 2 * r), rho, 1e-3, 31 + m)
    spr, ref, xr = _full_path(eng, X, F, r)
    np.testing.assert_allclose(spr.Sigma_r, ref['Sigma_r'], rtol=1e-8)
    np.testing.assert_array_equal(spr.sensors_, ref['piv'])
    assert spr.pivot_gap_.min() > 1e-9
    assert rel_fro(xr, ref['X_rec']) <= REL_FRO, (n_points, F, m, r)
    # a second placement through the masked entry (rows zeroed: the stored norms are dropped, the init sweep runs)
    mask = np.ones(X.shape[0], dtype=bool)
    mask[ref['piv'][:2]] = False
    spr.optimal_placement(mask=mask)
    U = ref['Ur'].copy()
    U[~mask] = 0.0
    np.testing.assert_array_equal(spr.sensors_, orc.qr_pivots(U)[0][:r])
print('SWITCH_CHECK_OK', {k: v for k, v in os.environ.items() if k.startswith('SPR_')})
