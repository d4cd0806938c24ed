"""Property tests (hypothesis) of OneHotRows against the dense ndarray it stands in for (reference :741-743): whatever the
reference's documentation does with C -- indexing, products, argmax, sums -- must give the same values, shapes and exception
types on both."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from openmeasure_amd.sparse_sensing import OneHotRows


@st.composite
def cases(draw):
    n = draw(st.integers(1, 40))
    s = draw(st.integers(1, 12))
    rows = draw(st.lists(st.integers(0, n - 1), min_size=s, max_size=s))
    return n, np.array(rows, dtype=np.int64)


def _dense(rows, n):
    C = np.zeros((len(rows), n))
    C[np.arange(len(rows)), rows] = 1.0
    return C


def _same(f_sparse, f_dense):
    try:
        want = f_dense()
    except Exception as e:                                   # noqa: BLE001 -- the exception TYPE is part of the contract
        with pytest.raises(type(e)):
            f_sparse()
        return
    got = f_sparse()
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)


@settings(max_examples=120, deadline=None)
@given(cases(), st.integers(-45, 45), st.integers(-45, 45))
def test_scalar_and_row_indexing(case, i, j):
    n, rows = case
    C, D = OneHotRows(rows, n), _dense(rows, n)
    _same(lambda: C[i, j], lambda: D[i, j])
    _same(lambda: C[i, :], lambda: D[i, :])
    _same(lambda: C[i], lambda: D[i])
    _same(lambda: np.argmax(C[i, :]), lambda: np.argmax(D[i, :]))
    _same(lambda: C[i, :][j], lambda: D[i, :][j])


@settings(max_examples=80, deadline=None)
@given(cases(), st.integers(1, 3), st.integers(0, 2 ** 31 - 1))
def test_products_and_reductions(case, k, seed):
    n, rows = case
    C, D = OneHotRows(rows, n), _dense(rows, n)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    Xk = rng.standard_normal((n, k))
    _same(lambda: C @ x, lambda: D @ x)
    _same(lambda: C.dot(x), lambda: D.dot(x))
    _same(lambda: C @ Xk, lambda: D @ Xk)
    _same(lambda: C @ x[:-1], lambda: D @ x[:-1]) if n > 1 else None
    _same(lambda: np.argmax(C, axis=1), lambda: np.argmax(D, axis=1))
    _same(lambda: np.argmax(C), lambda: np.argmax(D))
    _same(lambda: C.sum(), lambda: D.sum())
    _same(lambda: C.sum(axis=1), lambda: D.sum(axis=1))
    _same(lambda: C.sum(axis=0), lambda: D.sum(axis=0))
    assert C.shape == D.shape and len(C) == len(D) and C.ndim == D.ndim
    np.testing.assert_array_equal(np.asarray(C), D)
    np.testing.assert_array_equal(C.toarray(), D)
    np.testing.assert_array_equal(C.tocsr().toarray(), D)
    A = rng.standard_normal((2, len(rows)))
    _same(lambda: A @ C, lambda: A @ D)


@settings(max_examples=60, deadline=None)
@given(cases(), st.integers(-14, 14), st.integers(-14, 14))
def test_row_slices_stay_one_hot(case, a, b):
    n, rows = case
    C, D = OneHotRows(rows, n), _dense(rows, n)
    sub = C[a:b]
    assert isinstance(sub, OneHotRows)
    np.testing.assert_array_equal(np.asarray(sub), D[a:b]) if len(sub) else None
    assert sub.shape == D[a:b].shape
