"""GPU tests (-m gpu): bdrt_qp_box_batch -- the ridge / hyper-lambda ridge solve on the device (replaces
cvxopt.solvers.qp in Inverter._convex_opt, reference bayes_drt/inversion.py:1043-1067) -- against the host
implementation of the same interior-point algorithm, exact active-set answers (NNLS) and KKT residuals."""
import numpy as np
import pytest
from scipy.optimize import nnls

pytestmark = pytest.mark.gpu


def _batch(P, q, lo):
    from bayes_drt_amd import _lib
    from bayes_drt_amd._lib import ptr
    lib = _lib.require_gpu()
    P = np.ascontiguousarray(P, dtype=float); q = np.ascontiguousarray(q, dtype=float)
    nb, n = q.shape
    x = np.empty((nb, n)); obj = np.empty(nb); it = np.empty(nb, dtype=np.int32)
    lo = None if lo is None else np.ascontiguousarray(lo, dtype=float)
    rc = lib.bdrt_qp_box_batch(ptr(P), ptr(q), ptr(lo) if lo is not None else None, n, nb, ptr(x), ptr(obj), ptr(it))
    assert rc == 0, lib.bdrt_last_error()
    return x, obj, it


def _host(P, q, lo):
    from bayes_drt_amd import _lib
    from bayes_drt_amd._lib import ptr
    lib = _lib.load_library()
    n = len(q)
    x = np.empty(n); obj = np.zeros(1)
    rc = lib.bdrt_qp_box(ptr(np.ascontiguousarray(P)), ptr(np.ascontiguousarray(q)), ptr(np.ascontiguousarray(lo)), n, ptr(x), ptr(obj))
    assert rc >= 0
    return x, obj[0], rc


@pytest.mark.parametrize('n,m,nb', [(5, 12, 3), (40, 60, 7), (163, 162, 62), (230, 300, 2)])
def test_batched_gpu_qp_equals_host_solver_and_nnls(n, m, nb):
    """n = 163 x 62 problems is the Re-Im cross-validation shape (31 lambdas x 2 parts); n = 230 takes the global-memory
    factorisation (the packed KKT matrix no longer fits in LDS)."""
    rng = np.random.default_rng(n)
    P = np.empty((nb, n, n)); q = np.empty((nb, n))
    for b in range(nb):
        A = rng.standard_normal((m, n)); t = rng.standard_normal(m)
        lam = 10.0 ** rng.uniform(-6, 0)
        P[b] = A.T @ A + lam * np.eye(n); q[b] = -A.T @ t
    x, obj, it = _batch(P, q, np.zeros(n))
    assert np.all(x > -1e-7) and np.all(it > 0) and np.all(it < 80)      # infeasible-start method: x >= lo to feastol, like cvxopt
    for b in range(0, nb, max(1, nb // 5)):
        xh, oh, ith = _host(P[b], q[b], np.zeros(n))
        assert abs(obj[b] - oh) <= 1e-9 * max(1.0, abs(oh))
        assert np.max(np.abs(x[b] - xh)) <= 1e-6 * max(1.0, np.max(np.abs(xh)))
        assert abs(int(it[b]) - ith) <= 1
        g = P[b] @ x[b] + q[b]
        assert np.all(g > -1e-4) and np.sum(np.abs(g * x[b])) < 1e-5 * max(1.0, abs(obj[b]))   # KKT: duality gap <= reltol |obj|
    if n <= 40:
        for b in range(nb):
            lamI = P[b] - 0  # exact answer through NNLS on a Cholesky factor of P
            R = np.linalg.cholesky(P[b]).T
            xr, _ = nnls(R, np.linalg.solve(R.T, -q[b]))
            fr = 0.5 * xr @ P[b] @ xr + q[b] @ xr
            assert obj[b] - fr < 1e-6 * max(1.0, abs(fr)) and obj[b] >= fr - 1e-9


def test_mixed_and_free_bounds_on_the_gpu():
    rng = np.random.default_rng(1)
    n = 30
    A = rng.standard_normal((50, n)); t = rng.standard_normal(50) * 5
    P = (A.T @ A + 1e-2 * np.eye(n))[None]; q = (-A.T @ t)[None]
    lo = np.full(n, -10.0); lo[:2] = 0.0                   # nonneg=False in the reference (inversion.py:1060-1063)
    x, obj, it = _batch(P, q, lo)
    xh, oh, _ = _host(P[0], q[0], lo)
    assert np.all(x[0] > lo - 1e-6) and np.max(np.abs(x[0] - xh)) < 1e-6 * max(1.0, np.max(np.abs(xh)))
    xf, of, _ = _batch(P, q, None)                          # no bounds at all: one Newton step = the linear solve
    np.testing.assert_allclose(xf[0], np.linalg.solve(P[0], -q[0]), rtol=1e-8, atol=1e-10)


def test_semidefinite_P_is_regularised_like_the_host_solver():
    rng = np.random.default_rng(4)
    n = 20
    A = rng.standard_normal((8, n))                         # rank 8 < n
    P = (A.T @ A)[None]
    q = (A.T @ rng.standard_normal(8) + 0.05 * np.abs(rng.standard_normal(n)))[None]   # bounded below on x >= 0
    lo = np.zeros(n)
    x, obj, it = _batch(P, q, lo)
    xh, oh, _ = _host(P[0], q[0], lo)
    assert abs(obj[0] - oh) < 1e-7 * max(1.0, abs(oh)) and np.all(x > -1e-7)
