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
