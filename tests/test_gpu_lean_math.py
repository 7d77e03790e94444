"""GPU test (-m gpu): the lean exp / log / reciprocal of bdrt_device.h (what the one-block evaluator's inner loops use instead of
the device library's functions) against numpy, in units in the last place, over the ranges the evaluator feeds them."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(x):
    from bayes_drt_amd import _lib
    lib = _lib.require_gpu()
    fn = lib.bdrt_debug_lean_math
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; fn.restype = C.c_int
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty((3, x.size))
    assert fn(x.ctypes.data, x.size, out.ctypes.data) == 0
    return out


def _ulps(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_lean_exp():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-700, 700, 400000), rng.uniform(-40, 40, 400000), rng.uniform(-1, 1, 200000),
                        [0.0, -0.0, 1e-300, -1e-300, 709.7, -745.0, 0.5 * np.log(2.0), -0.5 * np.log(2.0)]])
    got = _run(x)[0]
    ref = np.exp(x.astype(np.longdouble))
    u = _ulps(got, ref.astype(np.float64))
    assert u.max() <= 1.5, (u.max(), x[np.argmax(u)])
    # beyond the finite range: inf and 0 (or a denormal), as the library's exp
    big = _run(np.array([710.0, 800.0, 1e5, -746.0, -1e5]))[0]
    assert np.all(np.isinf(big[:3])) and np.all(big[3:] == 0.0)
    assert np.isnan(_run(np.array([np.nan]))[0][0])


def test_lean_log():
    rng = np.random.default_rng(2)
    x = np.concatenate([np.exp(rng.uniform(-600, 600, 500000)), rng.uniform(0.5, 2.0, 400000), 1.0 + rng.uniform(-1e-6, 1e-6, 100000),
                        [1.0, 2.0, 0.5, np.sqrt(0.5), np.sqrt(2.0), 1e-300, 1e300]])
    got = _run(x)[1]
    ref = np.log(x.astype(np.longdouble)).astype(np.float64)
    bad = ref == 0.0
    assert np.all(got[bad] == 0.0)
    u = _ulps(got[~bad], ref[~bad])
    assert u.max() <= 1.5, (u.max(), x[~bad][np.argmax(u)])
    # outside the domain (zero, denormal, negative, inf, NaN): NaN, so that a point whose sigma^2 product has left the normal
    # range is rejected by the logarithm itself -- whatever the companion reciprocal returns
    out = _run(np.array([0.0, -0.0, 5e-324, 1e-310, -1.0, -1e-300, np.inf, -np.inf, np.nan]))[1]
    assert np.all(np.isnan(out)), out


def test_lean_rcp():
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(-650, 650, 600000)) * rng.choice([-1.0, 1.0], 600000), rng.uniform(0.5, 2.0, 400000), [1.0, -1.0, 3.0, 1e-300, 1e300]])
    got = _run(x)[2]
    ref = (np.longdouble(1.0) / x.astype(np.longdouble)).astype(np.float64)
    u = _ulps(got, ref)
    assert u.max() <= 1.0, (u.max(), x[np.argmax(u)])


def test_one_butterfly_for_eight_sums():
    """sum32_by_lane (bdrt_device.h): lane l of a half-wave ends with the total of quantity l & 7 (l & 3) over the 32 lanes -- against
    numpy, to rounding (another pairing than a sequential sum), on values of mixed sign and magnitude; the two halves are independent."""
    from bayes_drt_amd import _lib
    lib = _lib.require_gpu()
    fn = lib.bdrt_debug_sum_by_lane
    fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
    rng = np.random.default_rng(8)
    x = rng.standard_normal((64, 8)) * 10.0 ** rng.integers(-6, 7, (64, 8))
    out = np.empty((2, 64))
    assert fn(np.ascontiguousarray(x).ctypes.data, out.ctypes.data) == 0
    for half in (0, 1):
        rows = x[32 * half:32 * half + 32]
        tot, mag = rows.sum(axis=0), np.abs(rows).sum(axis=0)
        for l in range(32):
            assert abs(out[0, 32 * half + l] - tot[l & 7]) <= 1e-15 * 32 * mag[l & 7]
            assert abs(out[1, 32 * half + l] - tot[l & 3]) <= 1e-15 * 32 * mag[l & 3]
        # every lane that shares a quantity holds the same bits
        for j in range(8):
            assert len({out[0, 32 * half + l] for l in range(32) if l & 7 == j}) == 1


def test_leaf_acceptance_from_one_exponential_equals_the_textbook_form():
    """nuts_leaf_joins (bdrt_nuts_device.h): the device kernels take a leaf's new subtree log-weight and its acceptance decision
    u < exp(w - lsw_new) from ONE exponential, t = exp(-|lsw_sub - w|): lsw_new = max + log(1 + t) -- log_sum_exp2's own formula -- and
    exp(w - lsw_new) = (w >= lsw_sub ? 1 : t) / (1 + t).  Against the textbook form on the device (same lean exp / log): the log-weights
    agree (to the bit in all but a few cases per million), the decisions differ only where u meets the probability to rounding."""
    from bayes_drt_amd import _lib
    lib = _lib.require_gpu()
    fn = lib.bdrt_debug_leaf_joins
    fn.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5; fn.restype = C.c_int
    rng = np.random.default_rng(7)
    n = 1_000_000
    lsw = rng.uniform(-60.0, 20.0, n)
    lsw[: n // 50] = -np.inf                                    # the first leaf of a subtree
    w = np.where(rng.random(n) < 0.5, lsw + rng.normal(0.0, 3.0, n), rng.uniform(-80.0, 10.0, n))
    w[~np.isfinite(w)] = rng.uniform(-5.0, 5.0, int((~np.isfinite(w)).sum()))
    u = rng.random(n)
    # a share of the cases right at the boundary: u = the probability itself, and its neighbours
    m = np.maximum(lsw, w)
    p_np = np.exp(w - (m + np.log1p(np.exp(-np.abs(lsw - w)))))
    k = n // 10
    u[-k:] = np.clip(p_np[-k:] * (1.0 + rng.integers(-2, 3, k) * 2.0 ** -52), 0.0, 1.0)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    lsw, w, u = f64(lsw), f64(w), f64(u)
    ld, lr, pr = np.empty(n), np.empty(n), np.empty(n)
    jd, jr = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
    assert fn(lsw.ctypes.data, w.ctypes.data, u.ctypes.data, n, ld.ctypes.data, jd.ctypes.data, lr.ctypes.data, jr.ctypes.data,
              pr.ctypes.data) == 0
    ul = np.abs(ld - lr) / np.spacing(np.abs(lr))
    print('lsw_new: %d of %d differ, largest %.1f ulp; decisions differ in %d' % ((ul > 0).sum(), n, ul.max(), (jd != jr).sum()))
    # log_sum_exp2 itself, up to how the compiler contracts its last multiply-add at the two call sites: a handful of cases in a million,
    # where max + log(1 + t) cancels, differ in the last bits of the larger operand
    assert (ul > 0).sum() <= n // 10000 and np.all(np.abs(ld - lr) <= 2 * np.spacing(np.maximum(np.abs(lr), np.abs(m))))
    diff = jd != jr
    # where the two forms decide differently, u is within a few ulp of the probability (8 ulp: the lean exp's 1.5 ulp on either side
    # and the divisions)
    assert np.all(np.abs(u[diff] - pr[diff]) <= 8 * np.spacing(pr[diff])), (u[diff][:5], pr[diff][:5])
    assert diff.sum() <= k                                      # only among the planted boundary cases
    assert (jd[: n // 50] == 1).all()                           # an empty subtree always takes its first leaf
