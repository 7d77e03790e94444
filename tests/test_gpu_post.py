"""GPU tests (-m gpu) of the posterior post-processing kernels through the C ABI: column percentiles
and projected percentiles against numpy -- the functions the reference calls on the draws (np.percentile(..., axis=0),
reference bayes_drt/inversion.py:2560, :2702, :2734, :3068-3113)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QS = [0.0, 2.5, 25.0, 50.0, 75.0, 97.5, 100.0, 33.3333, 99.999]


@pytest.mark.parametrize('shape', [(1, 3), (2, 5), (7, 1), (1000, 37), (4096, 5), (5000, 11), (16384, 2)])
def test_percentile_is_numpy_percentile_bit_for_bit(shape):
    from bayes_drt_amd import post
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    X = rng.standard_normal(shape) * np.exp(rng.uniform(-20, 20, shape[1]))
    X[:, 0] = np.round(X[:, 0] / np.abs(X[:, 0]).max() * 4)          # heavy ties in one column
    got = post.percentile(X, QS, axis=0)
    want = np.percentile(X, QS, axis=0)
    assert got.shape == want.shape
    assert np.array_equal(got, want)                                  # same order statistics, same lerp formula
    assert post.percentile(X, 50.0).shape == (shape[1],)
    assert np.array_equal(post.percentile(X[:, -1], 97.5), np.percentile(X[:, -1], 97.5))   # 1-D samples, scalar q


def test_percentile_nan_column_and_range_checks():
    from bayes_drt_amd import post, _lib
    rng = np.random.default_rng(5)
    X = rng.standard_normal((300, 4))
    X[17, 2] = np.nan
    got = post.percentile(X, [10, 90])
    want = np.percentile(X, [10, 90], axis=0)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.all(np.isnan(got[:, 2]))
    assert np.array_equal(got[:, [0, 1, 3]], want[:, [0, 1, 3]])
    with pytest.raises(ValueError):
        post.percentile(X, 101.0)


def test_percentile_long_columns_match_numpy():
    """More draws than fit in LDS (np.percentile has no row limit; e.g. chains=4 x samples=5000): the HBM-scratch path."""
    from bayes_drt_amd import post
    rng = np.random.default_rng(6)
    X = rng.standard_normal((post.LDS_ROWS + 3617, 5)) * 3.0
    q = [0.0, 2.5, 50.0, 97.5, 100.0]
    assert np.array_equal(post.percentile(X, q), np.percentile(X, q, axis=0))


def test_projected_percentiles_match_numpy():
    """np.percentile(samples @ Phi.T + bias, q, axis=0): the draws-times-basis products of predict_Z / predict_distribution."""
    from bayes_drt_amd import post
    rng = np.random.default_rng(11)
    for rows, K, M in [(800, 163, 162), (37, 5, 3), (2000, 161, 200)]:
        X = np.exp(rng.standard_normal((rows, K)))
        Phi = rng.standard_normal((M, K))
        b = rng.standard_normal(M)
        got = post.project_percentile(X, Phi, b, QS)
        Y = X @ Phi.T + b
        want = np.percentile(Y, QS, axis=0)
        assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(Y))
        got0 = post.project_percentile(X, Phi, None, 50.0)
        assert np.max(np.abs(got0 - np.percentile(X @ Phi.T, 50.0, axis=0))) <= 1e-12 * np.max(np.abs(Y))


def test_sampler_percentiles_without_copying_the_draws():
    """bdrt_sampler_percentiles reduces the draws where they are (device); same numbers as numpy on the copied draws."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd._lib import NutsControl
    from tests.test_gpu_engine import _small_problem
    blk, Z, f, kw = _small_problem()
    prob = Problem([blk], Z, f, **kw)
    lib = prob._lib
    ctrl = NutsControl(); lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = 5
    n_units, warm, nd, D = 6, 20, 40, prob.D
    h = lib.bdrt_sampler_create(prob.handle, n_units, None, None, warm, nd, 77, None, C.byref(ctrl))
    assert h
    assert lib.bdrt_sampler_run(h) == 0
    draws = np.empty((n_units, nd, D)); lp = np.empty((n_units, nd))
    assert lib.bdrt_sampler_results(h, draws.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p), None) == 0
    q = np.array([2.5, 50.0, 97.5])
    lay = prob.layout()
    col0, K = lay['x'][0], prob.Ks[0]
    out = np.empty((3, K))
    rc = lib.bdrt_sampler_percentiles(h, 1, 5, col0, K, None, 0, None, q.ctypes.data_as(C.c_void_p), 3,
                                      out.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.bdrt_last_error()
    want = np.percentile(draws[1:5, :, col0:col0 + K].reshape(-1, K), q, axis=0)
    assert np.array_equal(out, want)
    # projected: percentiles of theta_x @ Phi^T
    Phi = np.random.default_rng(3).standard_normal((4, K))
    out2 = np.empty((3, 4))
    rc = lib.bdrt_sampler_percentiles(h, 0, n_units, col0, K, Phi.ctypes.data_as(C.c_void_p), 4, None,
                                      q.ctypes.data_as(C.c_void_p), 3, out2.ctypes.data_as(C.c_void_p))
    assert rc == 0
    Y = draws[:, :, col0:col0 + K].reshape(-1, K) @ Phi.T
    assert np.max(np.abs(out2 - np.percentile(Y, q, axis=0))) <= 1e-12 * np.max(np.abs(Y))
    assert lib.bdrt_sampler_percentiles(h, 0, n_units + 1, col0, K, None, 0, None, q.ctypes.data_as(C.c_void_p), 3,
                                        out.ctypes.data_as(C.c_void_p)) < 0
    lib.bdrt_sampler_destroy(h)
