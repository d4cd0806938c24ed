import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
_ALL = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))
GOLDEN = [n for n in _ALL if not n.startswith(('gem_', 'lim_', 'pinv_'))]       # fit -> qr placement -> train -> predict -> reconstruct
GOLDEN_GEM = [n for n in _ALL if n.startswith('gem_')]       # calc_type='gem' placement cases
GOLDEN_PINV = [n for n in _ALL if n.startswith('pinv_')]     # predict() on underdetermined / rank-deficient / ill-conditioned systems


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    g = {k: z[k] for k in z.files}
    g['n_features'] = int(g['n_features'])
    g['r'] = int(g['r'])
    g['select_modes'] = str(g['select_modes'])
    nm = float(g['n_modes'])
    g['n_modes'] = int(nm) if g['select_modes'] == 'number' else nm
    g['name'] = name
    g['scale_type'] = str(g['scale_type']) if 'scale_type' in g else 'std'
    g['axis_cnt'] = None if ('axis_cnt' in g and int(g['axis_cnt']) < 0) else 1
    return g


@pytest.fixture(params=GOLDEN)
def golden(request):
    return load_golden(request.param)


def load_golden_gem(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    g = {k: z[k] for k in z.files}
    for k in ('n_features', 'n_modes', 'n_sensors'):
        g[k] = int(g[k])
    g['d_min'] = float(g['d_min'])
    g['name'] = name
    return g


@pytest.fixture(params=GOLDEN_GEM)
def golden_gem(request):
    return load_golden_gem(request.param)


def load_golden_pinv(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    g = {k: z[k] for k in z.files}
    for k in ('n_features', 'n_modes'):
        g[k] = int(g[k])
    g['kind'] = str(g['kind'])
    g['name'] = name
    return g


@pytest.fixture(params=GOLDEN_PINV)
def golden_pinv(request):
    return load_golden_pinv(request.param)
