"""Posterior post-processing on the GPU: the reductions the reference applies to the HMC draws with
numpy right after `sampling` -- `np.percentile(samples, q, axis=0)` in `Inverter.coef_percentile`, `predict_Z`,
`predict_Rp`, `predict_sigma` (reference bayes_drt/inversion.py:2560, :2702, :2734, :3068-3113) and the
draws-times-basis products in front of them.  Same results as numpy's default ('linear') percentile."""
import ctypes as C

import numpy as np

from . import _lib

LDS_ROWS = 16384      # columns up to this many draws are sorted in LDS; longer ones in an HBM scratch (no limit)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def percentile(samples, q, axis=0):
    """np.percentile(samples, q, axis=0) on the GPU.  `samples`: [rows] or [rows x cols]; q scalar or sequence."""
    if axis != 0:
        raise ValueError('percentile: only axis=0 (over the draws) is implemented')
    return project_percentile(samples, None, None, q)


def project_percentile(samples, Phi, bias, q):
    """np.percentile(samples @ Phi.T + bias, q, axis=0) on the GPU (Phi [M x K]; Phi None: percentiles of `samples`)."""
    lib = _lib.require_gpu()
    X = np.asarray(samples, dtype=np.float64)
    one_d = X.ndim == 1
    if one_d:
        X = X[:, None]
    if X.ndim != 2:
        lead = X.shape[1:]
        X = X.reshape(X.shape[0], -1)
    else:
        lead = None
    X = np.ascontiguousarray(X)
    rows, K = X.shape
    qa = np.atleast_1d(np.asarray(q, dtype=np.float64))
    if not np.all((qa >= 0) & (qa <= 100)):              # (NaN included, like numpy)
        raise ValueError('Percentiles must be in the range [0, 100]')
    M = 0
    P = b = None
    if Phi is not None:
        P = np.ascontiguousarray(np.asarray(Phi, dtype=np.float64))
        if P.ndim != 2 or P.shape[1] != K:
            raise ValueError('project_percentile: Phi must be [M x %d]' % K)
        M = P.shape[0]
        if bias is not None:
            b = np.ascontiguousarray(np.asarray(bias, dtype=np.float64))
            if b.shape != (M,):
                raise ValueError('project_percentile: bias must be [%d]' % M)
    ncols = M if Phi is not None else K
    out = np.empty((qa.size, ncols))
    _lib.check(lib.bdrt_percentiles(_ptr(X), rows, K, K, _ptr(P), M, _ptr(b), _ptr(qa), qa.size, _ptr(out)),
               'bdrt_percentiles')
    if lead is not None and Phi is None:
        out = out.reshape((qa.size,) + lead)
    if one_d and Phi is None:
        out = out[:, 0]
    return out[0] if np.ndim(q) == 0 else out


def summary(draws, q, is_pos=None):
    """(mean [K], pct [len(q), K]) over the rows of `draws` [rows x K] after exp() of the columns flagged in `is_pos`
    (constrained scale of Stan <lower=0> parameters): np.mean / np.percentile of the reference (inversion.py:2517-2519,
    :2560) in one device pass -- the host-buffer form of bdrt_sampler_summary."""
    lib = _lib.require_gpu()
    X = np.ascontiguousarray(np.asarray(draws, dtype=np.float64))
    if X.ndim != 2:
        raise ValueError('summary: draws must be [rows x K]')
    rows, K = X.shape
    qa = np.ascontiguousarray(np.atleast_1d(np.asarray(q, dtype=np.float64)))
    if not np.all((qa >= 0) & (qa <= 100)):              # (NaN included, like numpy)
        raise ValueError('Percentiles must be in the range [0, 100]')
    mask = None
    if is_pos is not None:
        mask = np.ascontiguousarray(np.asarray(is_pos, dtype=np.uint8))
        if mask.shape != (K,):
            raise ValueError('summary: is_pos must have one entry per column')
    mean = np.empty(K); pct = np.empty((qa.size, K))
    _lib.check(lib.bdrt_summary(_ptr(X), rows, K, K, _ptr(mask), _ptr(qa), qa.size, _ptr(mean), _ptr(pct)), 'bdrt_summary')
    return mean, pct
