"""Problem = one Stan model instance resident in HBM (bdrt_problem of include/bdrt.h).

Mirrors the object the reference gets from `_get_stan_model` (bayes_drt/inversion.py:1566-1614) together with
the `dat` dict of `_prep_stan_data` (:1684-2122): `.optimizing(...)` and `.sampling(...)` live in engine.py.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Dat, check, f64, ptr


class Problem:
    """blocks: list of dicts {A [2nf x K], L0, L1, L2 [K x K] (mode-scaled), parallel, nonneg, x_scale};
    Z: [2nf] or [n_spectra x 2nf] stacked (Z', Z'')."""

    def __init__(self, blocks, Z, freq, sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0,
                 outlier_mode=0, so_lambda=10.0, so_alpha=5.0, so_beta=1.0, use_x_sum=None, x_sum_invscale=0.0):
        lib = _lib.require_gpu()
        self._lib = lib
        self._keep = []
        d = Dat()
        freq = f64(freq)
        Z = np.atleast_2d(f64(Z))
        d.nf = len(freq)
        if Z.shape[1] != 2 * d.nf:
            raise ValueError('Z must have 2*len(freq) columns (stacked real, imag)')
        d.nblocks = len(blocks)
        if not 1 <= d.nblocks <= _lib.MAXB:
            raise ValueError('1 to 3 distribution blocks are supported')
        for b, blk in enumerate(blocks):
            A = f64(blk['A'])
            if A.shape[0] != 2 * d.nf:
                raise ValueError('A must have 2*len(freq) rows (stacked real, imag)')
            K = A.shape[1]
            d.K[b] = K
            d.is_parallel[b] = int(bool(blk.get('parallel', False)))
            d.nonneg[b] = int(bool(blk.get('nonneg', False)))
            d.x_scale[b] = float(blk.get('x_scale', 1.0))
            arrs = [A] + [f64(blk[k]) for k in ('L0', 'L1', 'L2')]
            for L in arrs[1:]:
                if L.shape != (K, K):
                    raise ValueError('L matrices must be K x K')
            self._keep += arrs
            d.A[b], d.L0[b], d.L1[b], d.L2[b] = [a.ctypes.data for a in arrs]
        self._keep += [freq, Z]
        d.freq = freq.ctypes.data
        d.n_spectra = Z.shape[0]
        d.Z = Z.ctypes.data
        d.sigma_min, d.ups_alpha, d.ups_beta, d.induc_scale = sigma_min, ups_alpha, ups_beta, induc_scale
        d.outlier_mode = int(outlier_mode)
        d.so_lambda, d.so_alpha, d.so_beta = so_lambda, so_alpha, so_beta
        if use_x_sum is None:
            use_x_sum = len(blocks) > 1
        d.use_x_sum = int(bool(use_x_sum))
        d.x_sum_invscale = float(x_sum_invscale)
        self.dat = d
        self.nf = d.nf
        self.n_spectra = d.n_spectra
        self.Ks = [d.K[b] for b in range(d.nblocks)]
        self.handle = lib.bdrt_problem_create(C.byref(d))
        if not self.handle:
            raise _lib.BdrtError('bdrt_problem_create: ' + lib.bdrt_last_error().decode())
        self.D = check(lib.bdrt_num_params(self.handle), 'bdrt_num_params')
        pos = np.zeros(self.D, dtype=np.uint8)
        check(lib.bdrt_param_is_pos(self.handle, ptr(pos)), 'bdrt_param_is_pos')
        self.is_pos = pos.astype(bool)

    def evaluator(self):
        """Which tile evaluator the problem was given (bdrt_problem_evaluator, include/bdrt.h): 0 dense L, 1 banded L, 2 one-block
        fast tile, 3 general half-wave tile, 4 one-block fast tile with the A operands from the LDS-resident Toeplitz table."""
        return check(self._lib.bdrt_problem_evaluator(self.handle), 'bdrt_problem_evaluator')

    # layout of the unconstrained vector (Stan declaration order, include/bdrt.h)
    def layout(self):
        o = 2
        lay = {'x': [], 'ups': [], 'd': []}
        for K in self.Ks:
            lay['x'].append(o); o += K
        lay['err'] = o; o += 4
        lay['so'] = o
        if self.dat.outlier_mode:
            o += 2 * self.nf
        for K in self.Ks:
            lay['ups'].append(o); o += K
        for K in self.Ks:
            lay['d'].append(o); o += 3
        assert o == self.D
        return lay

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.bdrt_problem_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_Z(self, Z):
        Z = np.atleast_2d(f64(Z))
        check(self._lib.bdrt_problem_set_Z(self.handle, ptr(Z), Z.shape[0]), 'bdrt_problem_set_Z')
        self.n_spectra = Z.shape[0]

    def _spec(self, spec, B):
        if spec is None:
            return None
        s = np.ascontiguousarray(np.asarray(spec, dtype=np.int32))
        if s.shape != (B,):
            raise ValueError('spec must have one entry per row of theta')
        return s

    def logp_grad(self, theta, jacobian=True, spec=None, want_grad=True):
        theta = np.atleast_2d(f64(theta))
        B = theta.shape[0]
        if theta.shape[1] != self.D:
            raise ValueError('theta must have %d columns' % self.D)
        lp = np.empty(B)
        grad = np.empty((B, self.D)) if want_grad else None
        s = self._spec(spec, B)
        check(self._lib.bdrt_logp_grad(self.handle, ptr(theta), ptr(s), B, int(bool(jacobian)), ptr(lp), ptr(grad)),
              'bdrt_logp_grad')
        return lp, grad

    def transformed(self, theta, spec=None):
        theta = np.atleast_2d(f64(theta))
        B = theta.shape[0]
        params = np.empty((B, self.D)); Zh = np.empty((B, 2 * self.nf)); sg = np.empty((B, 2 * self.nf))
        s = self._spec(spec, B)
        check(self._lib.bdrt_transformed(self.handle, ptr(theta), ptr(s), B, ptr(params), ptr(Zh), ptr(sg)),
              'bdrt_transformed')
        return params, Zh, sg

    def constrain(self, theta):
        theta = f64(theta)
        return np.where(self.is_pos, np.exp(theta), theta)

    def unconstrain(self, params):
        params = f64(params)
        with np.errstate(invalid='ignore', divide='ignore'):
            return np.where(self.is_pos, np.log(np.where(self.is_pos, params, 1.0)), params)
