"""Host logic (no GPU): bdrt_qp_box, the interior-point replacement of cvxopt.solvers.qp in Inverter._convex_opt
(reference bayes_drt/inversion.py:1043-1067).  Checked by KKT residuals and against scipy's NNLS on a Cholesky
factor (an exact active-set answer) -- the interior point agrees to its tolerances and keeps strictly positive slacks."""
import numpy as np
from scipy.optimize import nnls

from bayes_drt_amd import _lib
from bayes_drt_amd._lib import ptr


def _qp(P, q, lo):
    lib = _lib.load_library()
    n = len(q)
    x = np.empty(n); obj = np.zeros(1)
    P = np.ascontiguousarray(P); q = np.ascontiguousarray(q); lo = np.ascontiguousarray(lo)
    rc = lib.bdrt_qp_box(ptr(P), ptr(q), ptr(lo), n, ptr(x), ptr(obj))
    assert rc >= 0, lib.bdrt_last_error()
    return x, obj[0], rc


def test_nonneg_qp_matches_nnls():
    rng = np.random.default_rng(0)
    for n, m in ((5, 12), (40, 60), (163, 162)):
        A = rng.standard_normal((m, n)); b = rng.standard_normal(m)
        lam = 1e-3
        P = A.T @ A + lam * np.eye(n); q = -A.T @ b
        x, obj, it = _qp(P, q, np.zeros(n))
        # exact solution: min |[A; sqrt(lam) I] x - [b; 0]| s.t. x >= 0
        xr, _ = nnls(np.vstack([A, np.sqrt(lam) * np.eye(n)]), np.concatenate([b, np.zeros(n)]))
        fr = 0.5 * xr @ P @ xr + q @ xr
        assert np.all(x > -1e-7)                               # cvxopt's method is infeasible-start: x >= lo only to feastol
        assert obj - fr < 1e-6 * max(1.0, abs(fr)) and obj >= fr - 1e-9
        assert np.max(np.abs(x - xr)) < 1e-3 * max(1.0, np.max(np.abs(xr)))
        # KKT: gradient g = Px + q >= 0 where x ~ 0, ~0 where x > 0
        g = P @ x + q
        assert np.all(g > -1e-5) and np.max(np.abs(g * x)) < 1e-5
        assert it < 60


def test_mixed_bounds_like_convex_opt():
    # nonneg=False in the reference: h = 10 except h[0:2] = 0  ->  x >= -10, x[0:2] >= 0 (inversion.py:1060-1063)
    rng = np.random.default_rng(1)
    n = 30
    A = rng.standard_normal((50, n)); b = rng.standard_normal(50) * 5
    P = A.T @ A + 1e-2 * np.eye(n); q = -A.T @ b
    lo = np.full(n, -10.0); lo[:2] = 0.0
    x, obj, it = _qp(P, q, lo)
    assert np.all(x > lo - 1e-6)
    g = P @ x + q                                                     # KKT: g >= 0, g (x - lo) = 0
    assert np.all(g > -1e-5) and np.sum(np.abs(g * (x - lo))) < 1e-5 * max(1.0, abs(obj))


def test_unbounded_variables_reduce_to_linear_solve():
    rng = np.random.default_rng(2)
    n = 12
    A = rng.standard_normal((20, n)); P = A.T @ A; q = rng.standard_normal(n)
    x, obj, it = _qp(P, q, np.full(n, -np.inf))
    np.testing.assert_allclose(x, np.linalg.solve(P, -q), rtol=1e-8, atol=1e-10)
