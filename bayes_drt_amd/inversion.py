"""Inverter -- the reference's user-facing API (bayes_drt/inversion.py, class Inverter) on the MI355X hot path.

Same constructor, `fit`, `ridge_fit`, `predict_*`, `coef_percentile`, `check_outliers` signatures, attribute names
and error behaviour as the reference, so that notebooks written for `bayes_drt.inversion.Inverter` run unchanged
with `from bayes_drt_amd.inversion import Inverter`.  What runs where:
  * A / L / M matrices          -> GPU (matrices.py -> bdrt_build_A/_L/_M)
  * `fit` MAP / HMC             -> GPU (stan_models.py -> engine.StanModel -> bdrt_optimize / bdrt_sampler_*)
  * `ridge_fit` Gram + QP       -> bdrt_gram (MFMA) + bdrt_qp_box_batch (interior point on the GPU, replaces cvxopt)
  * scaling, weights, Stan data dict, prediction algebra: numpy on the host (not hot: microseconds)
Out of scope (drift fits, MultiDist, fitY/SA, peak fitting, plotting, file loaders) raise
NotImplementedError instead of silently doing something else.
"""
import ctypes as C
import os
import warnings
from copy import deepcopy

import numpy as np

from . import _lib, post
from ._lib import f64, ptr
from .matrices import construct_A, construct_L, construct_M
from .stan_models import load_pickle
from .utils import get_outlier_thresh, is_loguniform, rel_round

script_dir = os.path.dirname(os.path.realpath(__file__))


def _same(a, b):
    try:
        np.testing.assert_equal(a, b)
        return True
    except AssertionError:
        return False


def _gaussian(y, epsilon):
    return np.exp(-(epsilon * y) ** 2)


class _QPResult(dict):
    """Mapping with the two keys the reference reads from cvxopt's result ('x', 'primal objective')."""



def _qp_batch(P, q, lo):
    """min 1/2 x'Px + q'x s.t. x >= lo for a stack of problems: one GPU launch (bdrt_qp.hip).

    A problem that reaches the iteration limit keeps its last iterate (cvxopt returns status 'unknown' with the best
    iterate in that case and the reference uses it, inversion.py:1066-1067); only a KKT matrix that is not positive
    definite is an error."""
    lib = _lib.require_gpu()
    P = np.ascontiguousarray(P, dtype=np.float64); q = np.ascontiguousarray(q, dtype=np.float64)
    nb, n = q.shape
    x = np.empty((nb, n)); obj = np.empty(nb)
    rc = lib.bdrt_qp_box_batch(ptr(P), ptr(q), ptr(np.ascontiguousarray(lo, dtype=np.float64)), n, nb, ptr(x), ptr(obj), None)
    if rc == -4:
        warnings.warn('bdrt_qp_box_batch: iteration limit reached; the last iterate is used')
    else:
        _lib.check(rc, 'bdrt_qp_box_batch')
    return x, obj


class _PhaseOffsets(object):
    """`correct_phase_offset` of ridge_fit (reference :302-338 initial estimate, :565-632 update inside the hyper-lambda loop):
    one constant phase offset per current-range (IERange) step of the instrument, fitted between two QP solves by
    minimising the phase residual against the current model plus an exponential prior on the offsets.  Frequencies are
    expected in descending order (instrument order), as the reference's index arithmetic assumes."""

    def __init__(self, frequencies, Z, IERange, lambda_phz, init_estimate):
        self.lambda_phz = lambda_phz
        self.phz_exp = np.angle(Z, deg=True)
        # steps of IERange, walking from low to high frequency; the last index closes the last segment
        self.steps = np.append(np.where(np.diff(np.asarray(IERange)[::-1]) != 0)[0] + 1, len(frequencies))
        self.offsets = np.zeros(len(self.steps))
        offset_vec = np.zeros(len(Z))
        adj = self.phz_exp.copy()[::-1]
        if init_estimate:
            for i, idx in enumerate(self.steps[:-1]):
                d = np.diff(adj)
                guess = adj[idx - 1] + (d[idx - 2] + d[idx]) / 2          # interpolated first difference across the step
                self.offsets[i] = guess - adj[idx]
                offset_vec[idx:self.steps[i + 1]] += self.offsets[i]
                adj[idx:self.steps[i + 1]] += self.offsets[i]
        self._store(Z, adj[::-1], offset_vec[::-1])
        self.Z = self.Z_adj.copy()                                        # what the loop treats as "the data" from here on

    def _store(self, Z, phz_adj, offset_vec):
        mod = np.abs(Z)
        self.phz_adj, self.offset_vec = phz_adj, offset_vec
        self.Z_adj = mod * np.cos(np.deg2rad(phz_adj)) + 1j * mod * np.sin(np.deg2rad(phz_adj))

    def _shifted(self, x):
        adj = self.phz_exp.copy()[::-1]
        vec = np.zeros(len(adj))
        for i, (idx, off) in enumerate(zip(self.steps[:-1], x)):
            vec[idx:self.steps[i + 1]] = off
            adj[idx:self.steps[i + 1]] += off
        return adj, vec

    def update(self, Z_pred):
        from scipy.optimize import minimize
        pred = np.angle(Z_pred, deg=True)
        var = np.var(np.angle(self.Z, deg=True) - pred)

        def cost(x):
            adj, _ = self._shifted(x)
            return 0.5 * np.sum((adj - pred[::-1]) ** 2) / var + self.lambda_phz * np.sum(np.abs(x))
        self.offsets = minimize(cost, x0=self.offsets)['x']
        adj, vec = self._shifted(self.offsets)
        self._store(self.Z, adj[::-1], vec[::-1])
        return self.Z_adj

    @staticmethod
    def scale_ratio(st, inv):
        """target_scaled / target of the reference (:628): the factor that takes an adjusted target to the fitted scale."""
        if not st['scale_Z']:
            return 1.0
        return 1.0 / inv._Z_scale if st['series'] else inv._Z_scale

    def snapshot(self):
        return {'phase_offsets': self.offsets.copy(), 'offset_vec': self.offset_vec.copy(), 'Zphz_adj': self.phz_adj.copy(),
                'Z_adj': self.Z_adj.copy()}


def _dist_equal(a, b):
    """Equality of two distribution-info dicts with array values (the reference's utils.check_equality)."""
    if set(a) != set(b):
        return False
    for k in a:
        va, vb = a[k], b[k]
        if isinstance(va, np.ndarray) or isinstance(vb, np.ndarray):
            if not (np.shape(va) == np.shape(vb) and np.allclose(va, vb, equal_nan=True)):
                return False
        elif va != vb:
            return False
    return True


# DDT options: allowed values and the defaults AS CODED in the reference (inversion.py:119), which differ from its
# docstring (:77 says transmissive).  The validation default of `bc` / `ct` below is what the reference's
# checks assume when the key is absent (:100-111).
_DDT_OPTIONS = (
    # key, allowed values, default used by the validity check, message
    ('dist_type', ('series', 'parallel'), 'parallel', "Invalid dist_type '{}' for distribution '{}'"),
    ('symmetry', ('planar', 'spherical'), 'planar', "Invalid symmetry '{}' for distribution '{}'"),
    ('bc', ('transmissive', 'blocking'), 'transmissive', "Invalid bc '{}' for distribution '{}'"),
    ('ct', (True, False), True, "Invalid ct {} for distribution '{}'"),
)
_DDT_DEFAULTS = {'dist_type': 'parallel', 'symmetry': 'planar', 'bc': 'blocking', 'ct': False}
_DRT_ONLY_INVALID = ('symmetry', 'bc', 'ct', 'k_ct')


def _invalidating_property(attr, check=None):
    """Inverter property whose assignment invalidates the cached matrices (reference :4069-4110)."""
    def getter(self):
        return getattr(self, attr)

    def setter(self, value):
        if check is not None:
            check(value)
        setattr(self, attr, value)
        self._invalidate()
    return getter, setter


def _check_basis(basis):
    if basis != 'gaussian':
        raise ValueError(f'Invalid basis {basis}. Options are gaussian')


def _validate_spectrum(frequencies, Z):
    """The reference hands NaN / inf / non-positive frequencies to cvxopt and Stan, which fail or return NaN in their own ways;
    here such input never reaches a kernel."""
    frequencies, Z = np.asarray(frequencies, dtype=float), np.asarray(Z)
    if len(frequencies) != len(Z):
        raise ValueError("Length of frequencies and Z must be equal")
    if not (np.all(np.isfinite(frequencies)) and np.all(np.isfinite(Z))):
        raise ValueError('frequencies and Z must be finite (NaN / inf found)')
    if np.any(frequencies <= 0):
        raise ValueError('frequencies must be positive')


def _check_basis_freq(basis_freq):
    # The reference assumes a descending basis (np.logspace(high, low, K)) without saying so: it never sorts basis_freq, and
    # with an ascending one its integrals over ln(tau) -- predict_Rp, the M matrices' spacing -- change sign.  Same here, so
    # say it once instead of returning a negative polarisation resistance silently.
    if basis_freq is not None and np.ndim(basis_freq) == 1 and len(basis_freq) > 1 and np.any(np.diff(np.asarray(basis_freq, dtype=float)) >= 0):
        warnings.warn('basis_freq is not in strictly descending order: like the reference, this class does not sort the basis, '
                      'and quantities integrated over ln(tau) (predict_Rp) change sign for an ascending one')


class _DeferredRidgeStart:
    """A ridge starting point of a MAP fit, set up but not solved yet (see `Inverter._resolve_deferred_ridge_starts`)."""
    def __init__(self, st, z_scale, frequencies, Z, nonneg, inductance_scale, max_iter):
        self.st, self.z_scale, self.frequencies, self.Z = st, z_scale, frequencies, Z
        self.nonneg, self.inductance_scale, self.max_iter = nonneg, inductance_scale, max_iter


class Inverter:
    def __init__(self, basis_freq=None, basis='gaussian', epsilon=None, fit_inductance=True,
                 distributions={'DRT': {'kernel': 'DRT'}}):
        """See the reference docstring (bayes_drt/inversion.py:29-49): basis_freq (10 points per decade recommended),
        basis ('gaussian' only), epsilon (None: 1/mean spacing of ln tau), fit_inductance (ridge_fit only),
        distributions {name: {kernel, dist_type, symmetry, bc, ct, k_ct, basis_freq, epsilon, x_scale}}."""
        self.distribution_matrices = {}
        self._invalidate()
        self.basis_freq, self.basis, self.epsilon = basis_freq, basis, epsilon
        self.fit_inductance = fit_inductance
        self.distributions = deepcopy(distributions)
        self._cached_distributions = self.distributions.copy()
        # state of "no fit yet" (reference :59-64)
        self.__dict__.update(f_train=[0], Z_train=None, _Z_scale=1.0, _init_params={}, distribution_fits={},
                             _iter_history=None)

    def _invalidate(self):
        """Matrices and prediction matrices must be rebuilt."""
        self._recalc_mat = True
        self.f_pred = None

    # ------------------------------------------------------------------ distributions (reference :66-137)
    def set_distributions(self, distributions):
        for name, info in distributions.items():
            kernel = info['kernel']
            if kernel == 'DRT':
                given = info.get('dist_type', 'series')
                if given != 'series':
                    warnings.warn("dist_type for DRT kernel must be series. Overwriting supplied dist_type '{}' for "
                                  "distribution '{}' with 'series'".format(name, given))
                info['dist_type'] = 'series'
                ignored = np.intersect1d(list(info.keys()), list(_DRT_ONLY_INVALID))
                if len(ignored) > 0:
                    warnings.warn("The following keys are invalid for distribution '{}': {}.\n These keys will be "
                                  "ignored".format(name, ignored))
            elif kernel == 'DDT':
                for key, allowed, assumed, msg in _DDT_OPTIONS:
                    if info.get(key, assumed) not in allowed:
                        raise ValueError(msg.format(info.get(key, 'NA'), name))
                if info.get('ct', False) == True and 'k_ct' not in info:
                    raise ValueError("k_ct must be supplied for distribution '{}' if ct==True".format(name))
                distributions[name] = dict(_DDT_DEFAULTS, **info)
            else:
                raise ValueError("Invalid kernel '{}' for distribution '{}'".format(kernel, name))
            self.distribution_matrices.setdefault(name, {})
        self._distributions = distributions
        self._invalidate()

    def get_distributions(self):
        return self._distributions

    distributions = property(get_distributions, set_distributions)

    # ------------------------------------------------------------------ properties (reference :4069-4110)
    get_basis_freq, set_basis_freq = _invalidating_property('_basis_freq', _check_basis_freq)
    basis_freq = property(get_basis_freq, set_basis_freq)
    get_basis, set_basis = _invalidating_property('_basis', _check_basis)
    basis = property(get_basis, set_basis)

    def get_epsilon(self):
        return self._epsilon

    def set_epsilon(self, epsilon, override_distributions=False):
        self._epsilon = epsilon
        self._invalidate()
        if override_distributions:
            for info in self.distributions.values():
                info['epsilon'] = epsilon

    epsilon = property(get_epsilon, set_epsilon)

    def get_fit_inductance(self):
        return self._fit_inductance_

    def set_fit_inductance(self, fit_inductance):
        self._fit_inductance_ = fit_inductance

    fit_inductance = property(get_fit_inductance, set_fit_inductance)

    # ================================================================== ridge (reference :142-1067)
    def ridge_fit(self, frequencies, Z, part='both', penalty='discrete', reg_ord=2, L1_penalty=0, scale_Z=True,
                  nonneg=True, weights=None, preset=None, hyper_lambda=True, hl_solution='analytic', hl_beta=2.5,
                  hl_fbeta=None, lambda_0=1e-2, cv_lambdas=np.logspace(-10, 5, 31), hyper_weights=False, hw_beta=2,
                  hw_wbar=1, xtol=1e-3, max_iter=20, hyper_a=False, alpha_a=2, hl_beta_a=2, hyper_b=False, sb=1,
                  correct_phase_offset=False, IERange=None, lambda_phz=1, init_phase_offset=False, x0=None, dZ=False,
                  dZ_power=0.5):
        """Hierarchical ridge fit of a single distribution (arguments as in the reference, :142-290).

        Where it runs: the Gram matrices (bdrt_gram) and every QP on the GPU; the standard fits -- ordinary ridge and the
        analytic hyper-lambda iteration, any penalty -- entirely on the GPU in one launch (bdrt_ridge); the rarely used
        variants that need scipy's scalar optimisers between two QPs (dZ weighting, hyper_a / hyper_b, hl_solution='lm',
        correct_phase_offset, hyper_weights) iterate on the host around the GPU QP."""
        if preset is not None:
            if preset not in ('Ciucci', 'Huang'):
                raise ValueError('Invalid preset {}. Options are {}'.format(preset, ['Ciucci', 'Huang']))
            if preset == 'Ciucci':
                penalty, lambda_0, hl_fbeta = 'discrete', 'cv', 0.1
            else:
                penalty, hl_beta, lambda_0, weights = 'integral', 2.5, 1e-2, 'modulus'
        if penalty in ('discrete', 'cholesky'):
            if np.min(hl_beta) <= 1:
                raise ValueError('hl_beta must be greater than 1 for penalty cholesky and discrete')
        elif penalty == 'integral':
            if np.min(hl_beta) <= 2:
                raise ValueError('hl_beta must be greater than 2 for penalty integral')
        else:
            raise ValueError(f'Invalid penalty argument {penalty}. Options are integral, discrete, and cholesky')
        if hyper_lambda and hyper_weights:
            raise ValueError('hyper_lambda and hyper_weights fits cannot be performed simultaneously')
        if len(self.distributions) > 1:
            raise ValueError('ridge_fit cannot be used to fit multiple distributions')
        if correct_phase_offset and IERange is None:
            raise ValueError('IERange must be provided if correct_phase_offset==True')
        if hl_solution not in ('analytic', 'lm'):
            raise ValueError("hl_solution must be 'analytic' or 'lm'")
        if part not in ('both', 'real', 'imag'):
            raise ValueError(f"Invalid part {part}. Options are 'both', 'real', or 'imag'")
        self.distribution_fits = {}
        Z = np.asarray(Z)
        phase = None
        if correct_phase_offset:
            phase = _PhaseOffsets(np.asarray(frequencies), Z, IERange, lambda_phz, init_phase_offset)
            Z = phase.Z_adj.copy()
        if isinstance(lambda_0, str) and lambda_0 == 'cv':
            lambda_0 = self.ridge_ReImCV(frequencies, Z, lambdas=cv_lambdas, penalty=penalty, hyper_lambda=hyper_lambda,
                                         hl_solution=hl_solution, hl_beta=hl_beta, hl_fbeta=hl_fbeta, reg_ord=reg_ord,
                                         L1_penalty=L1_penalty, x0=x0, weights=weights, xtol=xtol, max_iter=max_iter,
                                         scale_Z=scale_Z, nonneg=nonneg, dZ=dZ, dZ_power=dZ_power)
        st = self._ridge_setup(frequencies, Z, part, penalty, reg_ord, L1_penalty, scale_Z, nonneg, weights, dZ)
        device_loop = (not hyper_weights and not st['dZ'] and not hyper_a and not hyper_b and not correct_phase_offset
                       and hl_solution == 'analytic' and not os.environ.get('BDRT_HOST_LAMBDA_LOOP'))
        if device_loop:
            res = self._ridge_solve_device([st], [0], [lambda_0], hyper_lambda, hl_beta, hl_fbeta, x0, xtol, max_iter)[0]
        else:
            res = self._ridge_solve_host(st, lambda_0, hyper_lambda, hl_solution, hl_beta, hl_fbeta, hyper_weights, hw_beta,
                                         hw_wbar, xtol, max_iter, hyper_a, alpha_a, hl_beta_a, hyper_b, sb, phase, x0, dZ_power)
        self._ridge_finish(st, res, hyper_lambda, hyper_weights, max_iter)

    # ------------------------------------------------------------------ ridge: set-up shared by all solution paths
    def _ridge_setup(self, frequencies, Z, part, penalty, reg_ord, L1_penalty, scale_Z, nonneg, weights, dZ):
        """Matrices of one ridge problem (reference :370-470): augmented A, penalty bases, Gram matrix and q on the GPU."""
        name = list(self.distributions.keys())[0]
        info = self.distributions[name]
        if info['kernel'] != 'DRT' and dZ:
            warnings.warn('dZ should only be set to True for DRT recovery. Proceeding with dZ=False')
            dZ = False
        series = info['dist_type'] == 'series'
        target = Z if series else 1 / Z
        frequencies, target_s, WT_re, WT_im, W_re, W_im, dist_mat = self._prep_matrices(frequencies, target, part, weights,
                                                                                       dZ, scale_Z, penalty, 'ridge')
        info = self.distributions[name]
        if not series and scale_Z:
            # admittance target: the scale is defined on Z, the target is 1/Z_scaled (reference :378-383)
            target_s = 1 / self._scale_Z(1 / self.Z_train, 'ridge')
            WT_re, WT_im = W_re @ target_s.real, W_im @ target_s.imag
        mats = dist_mat[name]
        A_re, A_im, B = mats['A_re'], mats['A_im'], mats['B']
        K = A_re.shape[1]
        off = 2 if series else 0                    # augmented unknowns [R_inf, L/1e-4, x] for series (:402-417)
        n = K + off
        if series:
            Ar = np.zeros((len(frequencies), n)); Ar[:, 2:] = A_re; Ar[:, 0] = 1
            Ai = np.zeros((len(frequencies), n)); Ai[:, 2:] = A_im
            if self.fit_inductance:
                Ai[:, 1] = 2 * np.pi * frequencies * 1e-4
            A_re, A_im = Ar, Ai
            if B is not None:
                B = np.hstack((np.zeros((B.shape[0], 2)), B))

        def pad(Mx):
            out = np.zeros((n, n)); out[off:, off:] = Mx
            return out
        Ls = None
        if penalty in ('integral', 'cholesky'):
            base = [pad(mats['M%d' % o]) for o in (0, 1, 2)]
        if penalty in ('discrete', 'cholesky'):
            Ls = [np.hstack((np.zeros((K, off)), mats['L%d' % o])) for o in (0, 1, 2)]
        if penalty == 'discrete':
            base = [L.T @ L for L in Ls]
        if isinstance(reg_ord, (int, np.integer)):
            ro = np.zeros(3); ro[int(reg_ord)] = 1
            reg_ord = ro
        reg_ord = np.asarray(reg_ord, dtype=float)
        L1_vec = np.ones(n) * np.pi ** 0.5 / info['epsilon'] * L1_penalty
        if series:
            L1_vec[0:2] = 0
        lo = np.zeros(n) if nonneg else np.concatenate([np.zeros(min(2, n)), -10 * np.ones(n - min(2, n))])
        st = dict(name=name, info=info, series=series, part=part, penalty=penalty, reg_ord=reg_ord, L1_vec=L1_vec, lo=lo,
                  frequencies=frequencies, target_s=target_s, W_re=W_re, W_im=W_im, A_re=A_re, A_im=A_im, B=B, K=K, off=off,
                  n=n, base=base, Ls=Ls, scale_Z=scale_Z, dZ=dZ, tau=info['tau'],
                  # the reference zeroes the relative change of entry 1 (the inductance) in its convergence test whenever
                  # the inductance is not fitted -- for any distribution type (:733-734)
                  zero_delta1=bool(self.fit_inductance == False or part == 'real'))
        st['WT_re'], st['WT_im'] = WT_re, WT_im
        st['G'], st['g'] = self._ridge_gram(st, W_re, W_im, WT_re, WT_im)
        return st

    @staticmethod
    def _ridge_gram(st, W_re, W_im, T_re, T_im, part=None):
        """G = sum over the fitted parts of (W A)^T (W A), g = sum of (W A)^T (W T): on the GPU (bdrt_gram)."""
        rows, tg = [], []
        part = part or st['part']
        if part in ('both', 'real'):
            rows.append(W_re @ st['A_re']); tg.append(T_re)
        if part in ('both', 'imag'):
            rows.append(W_im @ st['A_im']); tg.append(T_im)
        WA = np.ascontiguousarray(np.vstack(rows)); WT = np.ascontiguousarray(np.concatenate(tg))
        lib = _lib.require_gpu()
        n = st['n']
        G = np.empty((n, n)); g = np.empty(n)
        _lib.check(lib.bdrt_gram(ptr(WA), ptr(WT), WA.shape[0], n, None, None, ptr(G), ptr(g)), 'bdrt_gram')
        return G, -g                                   # bdrt_gram returns q = -(WA^T WT)

    @classmethod
    def _unfitted_part_cost(cls, st, coef, W_re=None, W_im=None, T_re=None, T_im=None):
        """The reference's reported `cost` (fit dict, `_iter_history`) is 0.5 x^T P x + q^T x with P and q built from BOTH data
        parts whatever `part` is fitted (:711-713, :848-850), whereas the QP itself sees the fitted part only.  This is the
        unfitted part's share, added to the QP's own cost."""
        if st['part'] == 'both':
            return 0.0
        other = 'imag' if st['part'] == 'real' else 'real'
        if W_re is None:
            key = ('cost_other', other)
            if key not in st:
                st[key] = cls._ridge_gram(st, st['W_re'], st['W_im'], st['WT_re'], st['WT_im'], part=other)
            Go, go = st[key]
        else:
            Go, go = cls._ridge_gram(st, W_re, W_im, T_re, T_im, part=other)
        return 0.5 * coef @ Go @ coef - go @ coef

    @staticmethod
    def _hyper_prior_terms(penalty, hl_beta, lambda_0):
        """a, b of the gamma hyper-prior per derivative order and the implied lambda_0 / beta (reference :608-628)."""
        hl_beta = np.array([hl_beta] * 3, dtype=float) if np.ndim(hl_beta) == 0 else np.asarray(hl_beta, dtype=float)
        a = hl_beta / 2
        if penalty == 'integral':
            b = 0.5 * (2 * a - 2) / lambda_0
            lam0 = (2 * a - 2) / (2 * b)
        else:
            b = 0.5 * (2 * a - 1) / lambda_0
            lam0 = (2 * a - 1) / (2 * b)
        return a, b, lam0, 2 * a

    # ------------------------------------------------------------------ ridge: the whole fit on the GPU (bdrt_ridge)
    def _ridge_solve_device(self, setups, sel, lambdas, hyper_lambda, hl_beta, hl_fbeta, x0, xtol, max_iter):
        """A batch of ridge fits in ONE launch: fit j uses the data part setups[sel[j]] and lambda_0 = lambdas[j]."""
        lib = _lib.require_gpu()
        st0 = setups[0]
        n, K, off, nb, ng = st0['n'], st0['K'], st0['off'], len(sel), len(setups)
        o = _lib.RidgeOptions()
        o.n, o.K, o.off = n, K, off
        o.penalty = 1 if st0['penalty'] == 'integral' else 0
        # (bit g of zero_delta1: data part g -- the real-part fits of a cross-validation exclude the inductance from the
        #  convergence test, the imaginary-part fits do not)
        o.max_iter, o.hyper_lambda = int(max_iter), int(bool(hyper_lambda))
        o.zero_delta1 = sum(int(bool(s_['zero_delta1'])) << g for g, s_ in enumerate(setups))
        o.xtol = float(xtol)
        o.hl_fbeta = float(hl_fbeta) if (hl_fbeta is not None and st0['penalty'] != 'integral') else 0.0
        for i in range(3):
            o.reg_ord[i] = float(st0['reg_ord'][i])
        G = np.ascontiguousarray(np.stack([s['G'] for s in setups]))
        qbase = np.ascontiguousarray(np.stack([-s['g'] + s['L1_vec'] for s in setups]))
        base = np.ascontiguousarray(np.stack(st0['base']))
        Ls = np.ascontiguousarray(np.stack(st0['Ls'])) if st0['Ls'] is not None else None
        lam = np.ascontiguousarray(np.asarray(lambdas, dtype=np.float64))
        terms = [self._hyper_prior_terms(st0['penalty'], hl_beta, l) for l in lam]
        lam0s = np.ascontiguousarray(np.stack([t[2] for t in terms]))
        betas = np.ascontiguousarray(np.stack([t[3] for t in terms]))
        gsel = np.ascontiguousarray(np.asarray(sel, dtype=np.int32))
        x0a = None
        if x0 is not None:
            x0a = np.ascontiguousarray(np.broadcast_to(np.asarray(x0, dtype=np.float64), (nb, n)))
        coef = np.empty((nb, n)); lamv = np.empty((nb, 3, n)); cost = np.empty(nb); fun = np.empty(nb)
        iters = np.zeros(nb, dtype=np.int32); flags = np.zeros(nb, dtype=np.int32)
        mi = int(max_iter)
        hc = np.zeros((nb, mi, n)); hl = np.zeros((nb, mi, 3, n)); hf = np.zeros((nb, mi)); hk = np.zeros((nb, mi))
        rc = lib.bdrt_ridge(C.byref(o), nb, ng, ptr(G), ptr(qbase), ptr(gsel), ptr(base), ptr(Ls), ptr(f64(st0['lo'])), ptr(lam),
                            ptr(lam0s), ptr(betas), ptr(x0a), ptr(coef), ptr(lamv), ptr(cost), ptr(fun), ptr(iters), ptr(flags),
                            ptr(hc), ptr(hl), ptr(hf), ptr(hk))
        _lib.check(rc, 'bdrt_ridge')
        out = []
        for j in range(nb):
            if flags[j] & 4:
                warnings.warn('bdrt_ridge: a QP reached its iteration limit; the last iterate is used')
            sj = setups[sel[j]]
            hist = [{'lambda_vectors': [hl[j, t, i].copy() for i in range(3)], 'coef': hc[j, t].copy(), 'fun': hf[j, t],
                     'cost': hk[j, t] + self._unfitted_part_cost(sj, hc[j, t]),
                     'result': _QPResult({'x': hc[j, t].copy(), 'primal objective': hf[j, t]}),
                     'dZ_re': np.ones(n)} for t in range(int(iters[j]))]
            out.append(dict(coef=coef[j].copy(), lambda_vectors=[lamv[j, i].copy() for i in range(3)],
                            cost=cost[j] + self._unfitted_part_cost(sj, coef[j]),
                            result=_QPResult({'x': coef[j].copy(), 'primal objective': float(fun[j])}), history=hist,
                            converged=bool(flags[j] & 1), iterations=int(iters[j])))
        return out

    # ------------------------------------------------------------------ ridge: host iteration around the GPU QP (variants)
    def _ridge_solve_host(self, st, lambda_0, hyper_lambda, hl_solution, hl_beta, hl_fbeta, hyper_weights, hw_beta, hw_wbar,
                          xtol, max_iter, hyper_a, alpha_a, hl_beta_a, hyper_b, sb, phase, x0, dZ_power):
        n, off, K, series, penalty = st['n'], st['off'], st['K'], st['series'], st['penalty']
        base, Ls, reg_ord, lo, L1_vec = st['base'], st['Ls'], st['reg_ord'], st['lo'], st['L1_vec']
        A_re, A_im, B, tau = st['A_re'], st['A_im'], st['B'], st['tau']
        G, g = st['G'], st['g']

        def solve(G_, g_, L2_mat):
            """cvxopt.solvers.qp(P, q, -I, -lo) of the reference (:1043-1067): interior point on the GPU (bdrt_qp.hip)."""
            P = np.ascontiguousarray(G_ + L2_mat); q = np.ascontiguousarray(-g_ + L1_vec)
            x, obj = _qp_batch(P[None], q[None], lo)
            return _QPResult({'x': x[0], 'primal objective': float(obj[0])}), P, q

        def penalty_matrix(lams, dz):
            D = 1.0 / dz
            out = np.zeros((n, n))
            for Mb, lam, frac in zip(base, lams, reg_ord):
                if frac > 0:
                    sc = D * np.sqrt(lam)
                    out += frac * (sc[:, None] * Mb * sc[None, :])
            return out

        def converged(coef, prev):
            with np.errstate(divide='ignore', invalid='ignore'):
                delta = (coef - prev) / prev
            # the hyper-weights loop of the reference leaves the inductance out of the test only when it is not fitted
            # (:822: the `or part == 'real'` of the hyper-lambda loop, :733, is commented out there)
            if (self.fit_inductance == False) if hyper_weights else st['zero_delta1']:
                delta[1] = 0
            return np.mean(np.abs(delta)) < xtol

        dZ_re = np.ones(n)
        lam_vectors = [np.ones(n) * lambda_0 for _ in range(3)]
        history = []
        if hyper_lambda:
            a_list, b_list, lam0_list, beta_list = self._hyper_prior_terms(penalty, hl_beta, lambda_0)
            hyper_as = [np.ones(n) * a for a in a_list]
            hyper_bs = [np.ones(n) * b for b in b_list]
            lam0s = [np.ones(n) * l for l in lam0_list]
            betas = [np.ones(n) * bt for bt in beta_list]
            alpha_a = [alpha_a] * 3 if np.ndim(alpha_a) == 0 else list(alpha_a)
            hl_beta_a = [hl_beta_a] * 3 if np.ndim(hl_beta_a) == 0 else list(hl_beta_a)
            sb = [sb] * 3 if np.ndim(sb) == 0 else list(sb)
            hlb = np.array([hl_beta] * 3, dtype=float) if np.ndim(hl_beta) == 0 else np.asarray(hl_beta, dtype=float)
            coef = np.asarray(x0, dtype=float).copy() if x0 is not None else np.zeros(n) + 1e-6
            result, cost, ok, it = None, 0.0, False, 0
            while it < max_iter:
                prev = coef.copy()
                prev_lam = [l.copy() for l in lam_vectors]
                if st['dZ'] and it > 0:
                    dZ_raw = B @ prev / (np.mean(np.diff(np.log(tau))) / 0.23026)
                    dZ_re[off:] = np.abs(dZ_raw) ** dZ_power
                    dZ_re[np.abs(dZ_re < 1e-8)] = 1e-8      # (sic) the reference's mask, :527
                if hyper_b and it > 0:                      # _hyper_b (:985-990): b ~ normal(0, sb)
                    for i in range(3):
                        if reg_ord[i] > 0:
                            sl_, a_, s_ = np.sum(lam_vectors[i]), hyper_as[i], sb[i]
                            hb = 0.25 * (np.sqrt(16 * a_ * K * s_ ** 2 + 4 * s_ ** 4 * sl_ ** 2) - 2 * sl_ * s_ ** 2)
                            hb[hb < 1e-8] = 1e-8
                            hyper_bs[i] = hb
                            lam0s[i] = (2 * hyper_as[i] - 2) / hyper_bs[i]
                if hyper_a and it > 0:                      # _hyper_a (:992-1008): a - 1 ~ gamma(alpha_a, beta_a), scalar a
                    from scipy.optimize import minimize_scalar
                    from scipy.special import loggamma
                    for i in range(3):
                        if reg_ord[i] > 0:
                            slog = np.sum(np.log(hyper_bs[i] * lam_vectors[i]))
                            a_new = minimize_scalar(lambda a_: -2 * a_ * slog + 2 * loggamma(a_) + 2 * hl_beta_a[i] * (a_ - 1)
                                                    - 2 * (alpha_a[i] - 1) * np.log(a_ - 1), method='bounded', bounds=(1, 5))['x']
                            hyper_as[i] = np.ones(n) * a_new
                            lam0s[i] = (2 * hyper_as[i] - 2) / hyper_bs[i]
                            betas[i] = 2 * hyper_as[i]
                if phase is not None and it > 0:
                    Z_adj = phase.update(A_re @ prev + 1j * (A_im @ prev))
                    target_adj = (Z_adj if series else 1 / Z_adj) * phase.scale_ratio(st, self)
                    G, g = self._ridge_gram(st, st['W_re'], st['W_im'], st['W_re'] @ target_adj.real, st['W_im'] @ target_adj.imag)
                xs = prev / dZ_re
                for i in range(3):
                    if reg_ord[i] <= 0:
                        continue
                    if penalty in ('discrete', 'cholesky'):
                        if hl_solution == 'lm':             # :650-668
                            from scipy.optimize import least_squares
                            zeta = (hlb[i] - 1) / lambda_0
                            Lx2 = (Ls[i] @ prev) ** 2
                            r = least_squares(lambda x_: (Lx2 + zeta) * x_ - (hlb[i] - 1) * np.log(x_), prev_lam[i][off:],
                                              jac=lambda x_: np.diag(Lx2 + zeta - (hlb[i] - 1) / x_), method='lm',
                                              xtol=lambda_0 * 1e-3, max_nfev=100)
                            lam_vectors[i] = np.hstack((prev_lam[i][:off], r['x']))
                            continue
                        Lx2 = (Ls[i] @ xs) ** 2
                        if hl_fbeta is not None:           # _hyper_lambda_fbeta (:956-964)
                            lam = lambda_0 / (Lx2 / (np.max(Lx2) * hl_fbeta) + 1)
                        else:                              # _hyper_lambda_discrete (:947-954)
                            lam = 1 / (Lx2 / (betas[i][off:] - 1) + 1 / lam0s[i][off:])
                        lam_vectors[i] = np.hstack((np.ones(off), lam))
                    else:                                  # _hyper_lambda_integral (:973-983)
                        factor = (100, 10, 1)[i]
                        c = factor * xs
                        sl = np.sqrt(lam_vectors[i])
                        xlm = (c * sl)[:, None] * base[i] * c[None, :]
                        xlm = xlm - np.diag(np.diagonal(xlm))
                        Cv = np.sum(xlm, axis=0)
                        a = betas[i] / 2
                        b = 0.5 * (2 * a - 2) / lam0s[i]
                        d = c ** 2 * np.diagonal(base[i]) + 2 * b
                        lam = (Cv ** 2 - np.sign(Cv) * Cv * np.sqrt(4 * d * (2 * a - 2) + Cv ** 2) + 2 * d * (2 * a - 2)) / (2 * d ** 2)
                        lam[lam <= 0] = 1e-15
                        lam_vectors[i] = lam
                result, P, q = solve(G, g, penalty_matrix(lam_vectors, dZ_re))
                coef = np.array(result['x'])
                cost = 0.5 * coef @ P @ coef + q @ coef
                if phase is not None and it > 0:
                    cost += self._unfitted_part_cost(st, coef, st['W_re'], st['W_im'], st['W_re'] @ target_adj.real,
                                                     st['W_im'] @ target_adj.imag)
                else:
                    cost += self._unfitted_part_cost(st, coef)
                history.append({'lambda_vectors': [l.copy() for l in lam_vectors], 'coef': coef.copy(),
                                'fun': result['primal objective'], 'cost': cost, 'result': result, 'dZ_re': dZ_re.copy(),
                                'hyper_bs': [h.copy() for h in hyper_bs], 'hyper_lambda0s': [h.copy() for h in lam0s],
                                'hyper_hl_betas': [h.copy() for h in betas]})
                if phase is not None:
                    history[-1].update(phase.snapshot())
                it += 1
                if converged(coef, prev):
                    ok = True
                    break
            return dict(coef=coef, lambda_vectors=[l.copy() for l in lam_vectors], cost=cost, result=result, history=history,
                        converged=ok, iterations=it)
        if hyper_weights:
            coef = np.zeros(n) + 1e-6
            target_s = st['target_s']
            wbar = self._format_weights(st['frequencies'], target_s, hw_wbar, st['part'])
            w = wbar
            L2_mat = penalty_matrix(lam_vectors, dZ_re)
            result, cost, ok, it = None, 0.0, False, 0
            while it < max_iter:
                prev = coef.copy()
                if it > 0:                                 # _hyper_weights (:1010-1041)
                    zr, zi = hw_beta / np.real(wbar), hw_beta / np.imag(wbar)
                    res = target_s - (A_re @ coef + 1j * (A_im @ coef))
                    w = (np.real(wbar) - 1 / zr) / (res.real ** 2 / zr + 1) + 1j * (np.imag(wbar) - 1 / zi) / (res.imag ** 2 / zi + 1)
                Wr, Wi = np.diag(np.real(w)), np.diag(np.imag(w))
                G, g = self._ridge_gram(st, Wr, Wi, Wr @ target_s.real, Wi @ target_s.imag)
                result, P, q = solve(G, g, L2_mat)
                coef = np.array(result['x'])
                cost = 0.5 * coef @ P @ coef + q @ coef + self._unfitted_part_cost(st, coef, Wr, Wi, Wr @ target_s.real,
                                                                                    Wi @ target_s.imag)
                history.append({'weights': w.copy(), 'coef': coef.copy(), 'fun': result['primal objective'], 'cost': cost,
                                'result': result, 'dZ_re': dZ_re.copy()})
                it += 1
                if converged(coef, prev):
                    ok = True
                    break
            return dict(coef=coef, weights=w.copy(), cost=cost, result=result, history=history, converged=ok, iterations=it)
        result, P, q = solve(G, g, penalty_matrix(lam_vectors, dZ_re))
        coef = np.array(result['x'])
        return dict(coef=coef, cost=0.5 * coef @ P @ coef + q @ coef + self._unfitted_part_cost(st, coef), result=result,
                    history=None, converged=True, iterations=1)

    # ------------------------------------------------------------------ ridge: results -> attributes (reference :741-898)
    def _ridge_finish(self, st, res, hyper_lambda, hyper_weights, max_iter):
        name, info, series, part = st['name'], st['info'], st['series'], st['part']
        if hyper_lambda or hyper_weights:
            self._iter_history = res['history']
            if not res['converged'] and res['iterations'] >= max_iter:
                warnings.warn(f'Hyperparametric solution did not converge within {max_iter} iterations')
        fit = {'opt_result': res['result'], 'coef': res['coef'].copy(), 'cost': res['cost']}
        if hyper_lambda:
            fit['lambda_vectors'] = [l.copy() for l in res['lambda_vectors']]
        if hyper_weights:
            fit['weights'] = res['weights']
        self.distribution_fits[name] = fit
        fitc = fit['coef']
        target_s, A_re, A_im, frequencies = st['target_s'], st['A_re'], st['A_im'], st['frequencies']
        # the unfitted part's offset is recovered by least squares on the other part (:841-863); both are 1-D linear
        if part == 'imag' and series:
            fitc[0] = np.mean(target_s.real - A_re[:, 2:] @ fitc[2:])
        elif part == 'real' and series and self.fit_inductance:
            col = frequencies * 2 * np.pi * 1e-4
            fitc[1] = col @ (target_s.imag - A_im[:, 2:] @ fitc[2:]) / (col @ col)
        if st['scale_Z']:
            fit['scaled_coef'] = fitc.copy()
            fit['coef'] = self._rescale_coef(fitc, info['dist_type'])
        fitc = fit['coef']
        if series:
            fitc[1] *= 1e-4
            if not self.fit_inductance:
                fitc[1] = 0
            self.R_inf, self.inductance = fitc[0], fitc[1]
            fit['coef'] = fitc[2:]
        else:
            self.R_inf, self.inductance = 0, 0
        self.fit_type = 'ridge'

    def ridge_ReImCV(self, frequencies, Z, lambdas=np.logspace(-10, 5, 31), **kw):
        """Re-Im cross-validation for lambda_0 (reference :902-945).

        The reference runs the 2 x len(lambdas) hierarchical ridge fits one after the other.  They are independent and share
        their matrices, so here ALL of them are one launch of bdrt_ridge (one workgroup per fit, the hyper-lambda loop on the
        device).  Each fit performs exactly the arithmetic of a stand-alone `ridge_fit(part=..., lambda_0=...)`: same numbers
        as the sequential loop (`BDRT_SEQUENTIAL_CV=1` runs that loop).  Variants that iterate on the host (dZ, hl_solution='lm')
        take the sequential loop."""
        frequencies, Z = np.asarray(frequencies), np.asarray(Z)
        lambdas = np.asarray(lambdas, dtype=float)
        recv, imcv = np.zeros_like(lambdas), np.zeros_like(lambdas)
        kw = dict(kw)
        # what the one-launch path covers; any other ridge_fit keyword (hyper_weights, hyper_a, hyper_b, correct_phase_offset,
        # preset, ... -- the reference forwards **kw unchanged, :902-925) takes the sequential loop, and so does a mistyped
        # keyword: ridge_fit reports it
        defaults = dict(penalty='discrete', reg_ord=2, L1_penalty=0, scale_Z=True, nonneg=True, weights=None,
                        hyper_lambda=True, hl_beta=2.5, hl_fbeta=None, xtol=1e-3, max_iter=20, x0=None, dZ=False)
        host_variant = (kw.get('dZ', False) or kw.get('hl_solution', 'analytic') != 'analytic'
                        or bool(set(kw) - set(defaults) - {'hl_solution', 'dZ_power'}))
        if os.environ.get('BDRT_SEQUENTIAL_CV') or os.environ.get('BDRT_HOST_LAMBDA_LOOP') or host_variant:
            for i, lam in enumerate(lambdas):
                self.ridge_fit(frequencies, Z, part='real', lambda_0=lam, **kw)
                Zi = np.imag(self.predict_Z(frequencies))
                self.ridge_fit(frequencies, Z, part='imag', lambda_0=lam, **kw)
                Zr = np.real(self.predict_Z(frequencies))
                recv[i], imcv[i] = np.sum((Z.real - Zr) ** 2), np.sum((Z.imag - Zi) ** 2)
        else:
            o = dict(defaults, **{k: v for k, v in kw.items() if k in defaults})
            self.distribution_fits = {}
            setups = {}
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                for part in ('real', 'imag'):
                    setups[part] = self._ridge_setup(frequencies, Z, part, o['penalty'], o['reg_ord'], o['L1_penalty'],
                                                     o['scale_Z'], o['nonneg'], o['weights'], o['dZ'])
                jobs = [(i, part) for i in range(len(lambdas)) for part in ('real', 'imag')]
                res = self._ridge_solve_device([setups['real'], setups['imag']], [0 if p == 'real' else 1 for _, p in jobs],
                                               [lambdas[i] for i, _ in jobs], o['hyper_lambda'], o['hl_beta'], o['hl_fbeta'],
                                               o['x0'], o['xtol'], o['max_iter'])
                for (i, part), r in zip(jobs, res):
                    self.distribution_fits = {}
                    self._ridge_finish(setups[part], r, o['hyper_lambda'], False, o['max_iter'])
                    Zp = self.predict_Z(frequencies)
                    if part == 'real':
                        imcv[i] = np.sum((Z.imag - np.imag(Zp)) ** 2)
                    else:
                        recv[i] = np.sum((Z.real - np.real(Zp)) ** 2)
            # the reference leaves the last fit (imaginary part, last lambda) in place: so does the loop above
        tot = recv + imcv
        best = lambdas[np.argmin(tot)]
        if best == np.min(lambdas) or best == np.max(lambdas):
            warnings.warn('Optimal lambda_0 {} determined by Re-Im CV is at the boundary of the evaluated range. Re-run '
                          'with an expanded lambda_0 range to obtain an accurate estimate of the optimal lambda_0.'.format(best))
        self.cv_result = {'lambda': lambdas.copy(), 'recv': recv, 'imcv': imcv, 'totcv': tot}
        return best

    # ================================================================== Bayesian fit (reference :1072-1289)
    _RIDGE_SIDE_EFFECTS = ('distribution_fits', 'R_inf', 'inductance', 'fit_type', '_iter_history', 'cv_result', 'error_fit')

    def fit(self, frequencies, Z, part='both', scale_Z=True, nonneg=False, outliers=False, check_outliers=True,
            init_from_ridge=False, ridge_kw={}, sigma_min=0.002, inductance_scale=1, outlier_lambda=None,
            mode='optimize', random_seed=1234, max_iter=50000, warmup=200, samples=200, chains=2, add_stan_data={},
            model_str=None, fitY=False, SA=False, SASY=False, n_starts=None, algorithm=None):
        """Fit the distribution(s) with the calibrated hierarchical Bayesian model: mode='optimize' (MAP) or
        'sample' (NUTS).  Arguments as in the reference (:1072-1152), plus two that only concern mode='optimize':

        n_starts : None (default) -- the designated start (random, or the ridge solution with init_from_ridge=True) plus the
            other of the two for a single distribution, or plus three more random draws for several distributions; all
            starts are ONE lock-step batch on the GPU and the designated start's answer is kept unless another converged
            start ends at a clearly higher log-posterior (`_opt_report['start']`, `['starts']`).  This DIFFERS from the
            reference, which optimises from its single start only (:1216) and on multi-modal posteriors returns whatever
            mode that start leads to.  n_starts=1 is the reference's behaviour (also BDRT_MAP_SINGLE_START=1);
            n_starts=k > 1: the designated start, the other kind of start if there is one, then further random draws.
        algorithm : None / 'LBFGS+Newton' (default) -- the device-resident Newton iteration to a stationary point;
            'LBFGS' -- Stan's L-BFGS(5) with Stan's line search and termination tests alone, i.e. the kind of iterate the
            reference's `optimizing` call returns."""
        self._fit_argument_checks(part, mode, fitY, SA, SASY, n_starts, algorithm)
        job = self._fit_prepare(frequencies, Z, part, scale_Z, nonneg, outliers, init_from_ridge, ridge_kw, sigma_min,
                                inductance_scale, outlier_lambda, mode, add_stan_data, model_str, fitY, SA, SASY, n_starts)
        model, dat = job['model'], job['dat']
        if mode == 'optimize':
            self._opt_result = model.optimizing(dat, iter=max_iter, seed=random_seed, init=job['init'],
                                                extra_inits=job['extra_inits'], algorithm=algorithm or 'LBFGS+Newton')
            self._opt_report = model.last_report
        else:
            self._sample_result = model.sampling(dat, warmup=warmup, iter=warmup + samples, chains=chains,
                                                 seed=random_seed, init=job['init'], control=dict(self._NUTS_CONTROL))
        self._fit_finish(job, mode, sigma_min, check_outliers)

    _NUTS_CONTROL = {'adapt_delta': 0.9, 'adapt_t0': 10}        # reference :1221

    def _fit_prepare(self, frequencies, Z, part, scale_Z, nonneg, outliers, init_from_ridge, ridge_kw, sigma_min,
                     inductance_scale, outlier_lambda, mode, add_stan_data, model_str, fitY, SA, SASY, n_starts,
                     defer_ridge_start=False):
        """Everything `fit` does on the host before the engine runs (reference :1153-1214): initial values, matrices, model
        selection, the Stan data dict.  Sets the training attributes of this instance; returns the job description that
        `fit` hands to one engine call and `fit_many` to one batched call for all spectra.  `defer_ridge_start` (fit_many
        only): the ridge starting point of a MAP fit is set up here and solved with the other spectra's in one launch
        (`_resolve_deferred_ridge_starts`) -- an argument of this call, never a property of the instance, so that a later
        `fit` on a view that `fit_many` returned solves its own ridge start."""
        _validate_spectrum(frequencies, Z)
        init = 'random'
        if init_from_ridge:
            if len(self.distributions) > 1:
                raise ValueError('Ridge initialization can only be performed for single-distribution fits')
            init = self._get_init_from_ridge(frequencies, Z, mode, nonneg=nonneg, outliers=outliers,
                                             inductance_scale=inductance_scale, ridge_kw=ridge_kw)
            self._init_params = init()
        extra_inits = []
        if mode == 'optimize':
            extra_inits = self._map_extra_starts(n_starts, init_from_ridge, model_str, frequencies, Z, nonneg, outliers,
                                                 inductance_scale, ridge_kw, defer_ridge_start)
        frequencies, Z_scaled, dist_mat, outliers = self._bayes_matrices(frequencies, Z, part, scale_Z, outliers,
                                                                         init_from_ridge, ridge_kw)
        if model_str is None:
            model, model_str = self._get_stan_model(nonneg, outliers, False, None, fitY, SA)
        else:
            model = load_pickle(os.path.join(script_dir, 'stan_model_files', model_str))
        self.stan_model_name = model_str
        model_type = model_str.split('_')[0]
        if model_type == 'Series-Parallel' and nonneg == False:
            warnings.warn('For mixed series-parallel models, it is highly recommended to set nonnneg_drt=True')
        dat = self._prep_stan_data(frequencies, Z_scaled, part, model_type, dist_mat, outliers, sigma_min, mode=mode,
                                   inductance_scale=inductance_scale, outlier_lambda=outlier_lambda, fitY=fitY, SA=SA,
                                   SASY=SASY)
        dat.update(add_stan_data)
        if outliers and model_type == 'Series':
            dat['N'] = len(frequencies)        # package Series outlier models declare N = Nf (:1208-1211)
        self._stan_input = dat.copy()
        return dict(model=model, model_str=model_str, model_type=model_type, dat=dat, init=init, extra_inits=extra_inits,
                    outliers=outliers, frequencies=frequencies)

    def _fit_finish(self, job, mode, sigma_min, check_outliers):
        """Engine result (`_opt_result` / `_sample_result` of this instance) -> fit attributes, then the outlier check of
        the reference (:1223-1289)."""
        self._store_bayes_fit(job['model_type'], mode, sigma_min, job['outliers'])
        frequencies = job['frequencies']
        if job['outliers'] == False and check_outliers:
            idx = self.check_outliers(frequencies, self.Z_train, threshold=3.5, use_existing_fit=True)
            if len(idx) > 0:
                warnings.warn('Possible outliers were identified at indices {}, f={} Hz. Check the residuals and consider '
                              're-running with outliers=True'.format(idx, frequencies[idx]))

    def fit_many(self, frequencies, Z_list, part='both', scale_Z=True, nonneg=False, outliers=False, check_outliers=True,
                 init_from_ridge=False, ridge_kw={}, sigma_min=0.002, inductance_scale=1, outlier_lambda=None,
                 mode='optimize', random_seed=1234, max_iter=50000, warmup=200, samples=200, chains=2, add_stan_data={},
                 model_str=None, n_starts=None, algorithm=None, group=None):
        """The fits `[copy(self).fit(frequencies, Z, ...) for Z in Z_list]` -- the reference's own workload is such a loop
        over spectra measured on one frequency grid (code_EchemActa/Run fits.ipynb cells 4-5, inversion.py:1072-1081,
        :1218-1221) -- as ONE batch: one shared problem in HBM (the matrices are those of the common grid), every
        (spectrum, chain) unit of mode='sample' in one device-resident sampler, every (spectrum, start) fit of
        mode='optimize' in one lock-step batch.  Inside an initialised torch.distributed process group with more than one
        rank (or with `group` given) the units are sharded over the ranks' GPUs (`parallel.sample_sharded`: one broadcast
        of the problem, one scatter of the spectra, one gather of the results over RCCL); every rank returns the same list.

        Arguments as `fit`.  Each spectrum is scaled and weighted exactly as `fit` does it, and the random streams depend on
        (random_seed, chain) only: the result of spectrum i equals that of a separate `fit` call (same draws, same MAP).
        `frequencies` is one grid for all spectra or a list with one grid per spectrum (the reference's truncated-spectrum
        study, Run fits.ipynb cells 13-14: spectra of 53 and 91 frequencies through one loop); `sigma_min`, `nonneg`, `outliers`,
        `inductance_scale` and `outlier_lambda` take one value or a list with one per spectrum (the same study: sigma_min = 0.005
        for the noiseless spectrum, 0.002 otherwise).  Spectra that share grid, basis, model and options are one batch -- one
        problem in HBM, one engine call --, the others further batches of the same call; the list comes back in input order.
        Returns a list of Inverter objects (shallow copies of this one) carrying the fit attributes of `fit`, readable
        through `predict_*`, `coef_percentile`, ...; this instance itself is left as it was."""
        self._fit_argument_checks(part, mode, False, False, False, n_starts, algorithm)
        views, jobs = self._batch_jobs(frequencies, Z_list, part, scale_Z, nonneg, outliers, init_from_ridge, ridge_kw, sigma_min,
                                       inductance_scale, outlier_lambda, mode, add_stan_data, model_str, n_starts)
        if not views:
            return []
        # one engine call per group of spectra that share everything but Z: the model (outliers='auto' may choose differently per
        # spectrum), the frequency grid and basis (hence the matrices) and every Stan data entry the options decide; results stay
        # in input order
        order = {}
        for i, job in enumerate(jobs):
            order.setdefault(self._batch_key(job), []).append(i)
        for idx in order.values():
            self._fit_batch([views[i] for i in idx], [jobs[i] for i in idx], mode, random_seed, max_iter, warmup, samples,
                            chains, algorithm, group)
        for inv, job in zip(views, jobs):
            inv._fit_finish(job, mode, job['sigma_min'], check_outliers)
        return views

    @staticmethod
    def _batch_key(job):
        """Spectra with equal keys share one problem in HBM: the model file, the frequency grid (hence, on this instance's basis,
        the matrices) and the options that enter the Stan data apart from 'Z' (`_stack_stan_data` verifies the entries)."""
        return job['batch_key']

    @staticmethod
    def _per_spectrum(value, n, name):
        """An option of `fit_many` given once (every spectrum) or as a list / array of one value per spectrum."""
        if isinstance(value, (list, tuple, np.ndarray)):
            if len(value) != n:
                raise ValueError('fit_many: %s has %d entries for %d spectra' % (name, len(value), n))
            return list(value)
        return [value] * n

    def _batch_jobs(self, frequencies, Z_list, part, scale_Z, nonneg, outliers, init_from_ridge, ridge_kw, sigma_min,
                    inductance_scale, outlier_lambda, mode, add_stan_data, model_str, n_starts):
        """`_fit_prepare` for every spectrum of a batch, each on a shallow copy of this instance.  The copies of one frequency
        grid share a matrix cache (nothing is built twice); every grid has a cache of its own, seeded from this instance's and
        detached from it, so that neither this instance nor the views of another grid ever see matrices of a grid they were
        not built for.  `frequencies`: one grid for all spectra, or a list with one grid per spectrum; `sigma_min`, `nonneg`,
        `outliers`, `inductance_scale`, `outlier_lambda`: one value, or a list with one per spectrum.  Returns (views, jobs)."""
        import copy
        Z_list = [np.asarray(Z) for Z in Z_list]
        n = len(Z_list)
        per_grid = isinstance(frequencies, (list, tuple)) and n > 0 and len(frequencies) == n and \
            all(np.ndim(f) == 1 for f in frequencies)
        if isinstance(frequencies, np.ndarray) and frequencies.ndim == 2:
            per_grid = True
            if len(frequencies) != n:
                raise ValueError('fit_many: %d frequency grids for %d spectra' % (len(frequencies), n))
        freqs = [np.asarray(f, dtype=float) for f in frequencies] if per_grid else [frequencies] * n
        opt = {k: self._per_spectrum(v, n, k) for k, v in (('sigma_min', sigma_min), ('nonneg', nonneg), ('outliers', outliers),
                                                          ('inductance_scale', inductance_scale),
                                                          ('outlier_lambda', outlier_lambda))}
        views, jobs, last_of_grid = [], [], {}
        for i, (f, Z) in enumerate(zip(freqs, Z_list)):
            fa = np.asarray(f, dtype=float)
            key = rel_round(np.sort(fa)[::-1], 10).tobytes() if fa.ndim == 1 and len(fa) > 1 and np.all(np.isfinite(fa)) and \
                np.all(fa > 0) else None
            base = last_of_grid.get(key)
            if base is None:
                inv = copy.copy(self)
                inv.distribution_matrices = {name: dict(st, **({'_penalty': dict(st['_penalty'])} if '_penalty' in st else {}))
                                             for name, st in self.distribution_matrices.items()}
            else:
                inv = copy.copy(base)
            job = inv._fit_prepare(f, Z, part, scale_Z, opt['nonneg'][i], opt['outliers'][i], init_from_ridge, ridge_kw,
                                   opt['sigma_min'][i], opt['inductance_scale'][i], opt['outlier_lambda'][i], mode, add_stan_data,
                                   model_str, False, False, False, n_starts, defer_ridge_start=(mode == 'optimize'))
            job['sigma_min'] = opt['sigma_min'][i]
            job['batch_key'] = (job['model_str'], key, repr(opt['sigma_min'][i]), repr(opt['inductance_scale'][i]),
                                repr(opt['outlier_lambda'][i]))
            views.append(inv); jobs.append(job)
            if key is not None:
                last_of_grid[key] = inv
        return views, jobs

    @staticmethod
    def _stack_stan_data(jobs):
        """One Stan data dict for spectra that share everything but Z: 'Z' becomes [n_spectra, 2 Nf]."""
        d0 = jobs[0]['dat']
        for job in jobs[1:]:
            if set(job['dat']) != set(d0):
                raise ValueError('the spectra of one batch do not share one model')
            for k, v in job['dat'].items():
                if k != 'Z' and not np.array_equal(np.asarray(v), np.asarray(d0[k])):
                    raise ValueError('the spectra of one batch do not share the Stan data entry %r' % k)
        dat = dict(d0)
        dat['Z'] = np.vstack([np.asarray(job['dat']['Z'], dtype=float).reshape(1, -1) for job in jobs])
        return dat

    def batch_stan_data(self, frequencies, Z_list, part='both', scale_Z=True, nonneg=False, outliers=False, sigma_min=0.002,
                        inductance_scale=1, outlier_lambda=None, mode='sample', add_stan_data={}):
        """(model file name, Stan data dict with 'Z' = [n_spectra, 2 Nf]) of the batch `fit_many` would run: every spectrum scaled
        and weighted as `fit` does it (reference :1153-1214).  For callers that drive the engine themselves (bench.py)."""
        _, jobs = self._batch_jobs(frequencies, Z_list, part, scale_Z, nonneg, outliers, False, {}, sigma_min, inductance_scale,
                                   outlier_lambda, mode, add_stan_data, None, 1)
        if len({self._batch_key(job) for job in jobs}) != 1:
            raise ValueError('batch_stan_data: the spectra do not share one model, one grid and one set of options')
        return jobs[0]['model_str'], self._stack_stan_data(jobs)

    @staticmethod
    def _fit_batch(views, jobs, mode, random_seed, max_iter, warmup, samples, chains, algorithm, group):
        """One engine call for the spectra of one model: fills `_opt_result` / `_sample_result` of every view."""
        import ctypes as C
        from . import engine, parallel
        from ._lib import NutsControl
        ns = len(jobs)
        dat = Inverter._stack_stan_data(jobs)
        model = jobs[0]['model']
        P = model._prepare(dat)
        world = 1
        dist = None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                world = dist.get_world_size(group)
        except ImportError:
            pass
        if mode == 'optimize':
            Inverter._resolve_deferred_ridge_starts(views, jobs)
            # rows: for each spectrum its designated start, then its extra starts -- the rows `optimizing` builds for one fit
            rows, owner = [], []
            for i, job in enumerate(jobs):
                r = [model._init_theta(job['init'], 1, random_seed)]
                for e in job['extra_inits']:
                    try:
                        if isinstance(e, tuple) and e[0] == 'random':
                            r.append(model._init_theta('random', 1, random_seed + int(e[1])))
                        else:
                            r.append(model._init_theta(e, 1, random_seed))
                    except ValueError:
                        pass
                rows += r; owner += [i] * len(r)
            theta0 = np.vstack(rows)
            owner = np.asarray(owner, dtype=np.int32)
            opts = {}
            alg = algorithm or 'LBFGS+Newton'
            if alg == 'LBFGS':
                opts['newton_max_iter'] = 0
            if world > 1:
                rank = dist.get_rank(group)
                lo, hi = parallel.shard_bounds(len(owner), world, rank)
                out_l, rep_l = (engine.optimize_batch(P, theta0[lo:hi], spec=owner[lo:hi], max_iter=max_iter, **opts)
                                if hi > lo else (np.zeros((0, P.D)), []))
                keys = ('iterations', 'n_evals', 'return_code', 'lp', 'grad_norm', 'newton_iterations', 'grad_inf')
                loc = np.array([[r[k] for k in keys] for r in rep_l], dtype=float).reshape(hi - lo, len(keys))
                counts = [b - a for a, b in (parallel.shard_bounds(len(owner), world, r) for r in range(world))]
                out = parallel.gather_rows(out_l, counts, group)
                rep = parallel.gather_rows(loc, counts, group)
                reports = [dict((k, (int(v) if k in ('iterations', 'n_evals', 'return_code', 'newton_iterations') else float(v)))
                                for k, v in zip(keys, row)) for row in rep]
            else:
                out, reports = engine.optimize_batch(P, theta0, spec=owner, max_iter=max_iter, **opts)
            for i, inv in enumerate(views):
                sel = np.nonzero(owner == i)[0]
                best = 0
                for j in range(1, len(sel)):                       # the rule of StanModel.optimizing
                    a, b = reports[sel[j]], reports[sel[best]]
                    higher = np.isfinite(a['lp']) and (not np.isfinite(b['lp']) or a['lp'] > b['lp'] + 1e-6 * max(1.0, abs(b['lp'])))
                    if higher and (a['return_code'] == 0 or b['return_code'] != 0):
                        best = j
                inv._opt_result = model.result_dict(out[sel[best]])
                inv._opt_report = dict(reports[sel[best]], start=best, starts=[reports[k] for k in sel])
            return
        # ---- mode == 'sample'
        n_draws = int(samples)
        spec, chain = parallel.make_units(ns, chains)
        init_theta = None
        if any(not (isinstance(job['init'], str) and job['init'] == 'random') for job in jobs):
            init_theta = np.vstack([model._init_theta(job['init'], chains, random_seed) for job in jobs])
        control = dict(Inverter._NUTS_CONTROL)
        if world > 1:
            blocks, kw, _ = engine.blocks_from_dat(model.model_name, dat)
            pk = dict(kw, blocks=blocks, Z=np.asarray(dat['Z'], dtype=float), freq=np.asarray(dat['freq'], dtype=float))
            res = parallel.sample_sharded(pk if dist.get_rank(group) == 0 else None, ns, chains, warmup, n_draws, seed=random_seed,
                                          control=control, group=group, gather='draws', init_theta=init_theta)
            draws, lp = res['draws'], res['lp']
            diag = [dict(n_leapfrog=int(r[0]), n_divergent=int(r[1]), n_max_treedepth=int(r[2]), stepsize=float(r[3]),
                         mean_accept=float(r[4])) for r in res['stats']]
        else:
            ctrl = NutsControl()
            P._lib.bdrt_nuts_defaults(C.byref(ctrl))
            for k, v in control.items():
                setattr(ctrl, k, v)
            draws, lp, diag = engine.sample_units(P, ns * chains, warmup, n_draws, random_seed, ctrl, spec=spec, chain_ids=chain,
                                                  init_theta=init_theta)
        for i, inv in enumerate(views):
            u0, u1 = i * chains, (i + 1) * chains
            inv._sample_result = engine.StanFit(model, draws[u0:u1].reshape(chains * n_draws, P.D), lp[u0:u1].reshape(-1),
                                                diag[u0:u1], chains, n_draws)

    @staticmethod
    def _resolve_deferred_ridge_starts(views, jobs):
        """The ridge starting points that `_map_extra_starts` set up without solving (fit_many): all of them in one launch of
        bdrt_ridge -- a workgroup per spectrum instead of a 2.4 ms launch per spectrum (61 of them were 145 ms of the published MAP
        study's 0.5 s) --, then per spectrum what `_get_init_from_ridge` does with a finished ridge fit.  Each solve performs the
        arithmetic of the stand-alone call (`_ridge_solve_device`), so the starting points are those of separate `fit` calls."""
        pend = [(inv, job, k) for inv, job in zip(views, jobs) for k, e in enumerate(job['extra_inits']) if isinstance(e, _DeferredRidgeStart)]
        if not pend:
            return
        d0 = pend[0][1]['extra_inits'][pend[0][2]]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = pend[0][0]._ridge_solve_device([job['extra_inits'][k].st for _, job, k in pend], list(range(len(pend))), [1] * len(pend),
                                                 True, 5, None, None, 1e-3, d0.max_iter)
            for (inv, job, k), r in zip(pend, res):
                d = job['extra_inits'][k]
                keys = inv._RIDGE_SIDE_EFFECTS + ('_Z_scale',)
                saved = {a: getattr(inv, a) for a in keys if hasattr(inv, a)}
                try:
                    inv.distribution_fits = {}
                    inv._Z_scale = d.z_scale
                    inv._ridge_finish(d.st, r, True, False, d.max_iter)
                    job['extra_inits'][k] = inv._init_values_from_ridge_fit(d.frequencies, d.Z, d.nonneg, False, d.inductance_scale)
                except Exception as e:                   # the ridge candidate is optional (as in `_map_extra_starts`)
                    warnings.warn('ridge starting point not available (%s): MAP from the random start only' % e)
                    job['extra_inits'][k] = None
                finally:
                    for a in keys:
                        if a in saved:
                            setattr(inv, a, saved[a])
                        elif hasattr(inv, a):
                            delattr(inv, a)
        for _, job, _k in pend:
            job['extra_inits'] = [e for e in job['extra_inits'] if e is not None]

    @staticmethod
    def _fit_argument_checks(part, mode, fitY, SA, SASY, n_starts, algorithm):
        if fitY or SA or SASY:
            raise NotImplementedError('fitY / SA / SASY are experimental flags of the reference ("for testing only") and '
                                      'are not part of this build')
        if part not in ('both', 'real', 'imag'):
            raise ValueError(f"Invalid part {part}. Options are 'both', 'real', or 'imag'")
        if mode not in ('optimize', 'sample'):
            raise ValueError("mode must be 'optimize' or 'sample'")
        if n_starts is not None and (not isinstance(n_starts, (int, np.integer)) or n_starts < 1):
            raise ValueError('n_starts must be a positive integer or None')
        if algorithm not in (None, 'LBFGS', 'LBFGS+Newton'):
            raise ValueError("algorithm must be None, 'LBFGS' or 'LBFGS+Newton'")

    def _map_extra_starts(self, n_starts, init_from_ridge, model_str, frequencies, Z, nonneg, outliers, inductance_scale, ridge_kw,
                          defer_ridge_start=False):
        """Further MAP starting points next to the designated one (see `fit`).  The hierarchical posterior has poor local
        maxima -- everything explained as noise, or a huge Z_hat with a proportionally huge error -- that a single random
        start reaches on sparse or outlier-ridden spectra; the starts run as one batch in lock-step on the GPU, the
        ridge candidate costs its (shortened) ridge solve."""
        if n_starts is None and os.environ.get('BDRT_MAP_SINGLE_START'):
            n_starts = 1
        if n_starts == 1:
            return []
        single = len(self.distributions) == 1 and model_str is None
        extra = []
        if init_from_ridge:
            extra.append('random')                       # the ridge start stays the designated one
        elif single:
            # the under-fitted ridge solution as second start; ridge_fit overwrites fit attributes on the way, which a
            # Bayesian fit would not touch: put them back
            saved = {k: getattr(self, k) for k in self._RIDGE_SIDE_EFFECTS if hasattr(self, k)}
            try:
                # (two hyper-lambda iterations, not the twenty of a ridge fit in its own right: this is a starting point, and
                #  the full ridge solve -- 19 ms at K = 81, 57 ms at K = 161, one workgroup -- was 40 % of the whole MAP fit; with two
                #  the start is as good as with three -- 49 against 54 Newton rounds at K = 161 --, with one it is not: 136)
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    if defer_ridge_start and not ridge_kw and not outliers and \
                            not os.environ.get('BDRT_HOST_LAMBDA_LOOP'):
                        # fit_many: set the ridge problem up now, solve it with the other spectra's in ONE launch of bdrt_ridge
                        # right before the batch of MAP fits starts (`_resolve_deferred_ridge_starts`); same arguments as the
                        # ridge_fit call of `_get_init_from_ridge` makes
                        self.distribution_fits = {}
                        st = self._ridge_setup(frequencies, np.asarray(Z), 'both', 'integral', 2, 0, True, True, 'modulus', False)
                        extra.append(_DeferredRidgeStart(st, self._Z_scale, frequencies, Z, nonneg, inductance_scale,
                                                         int(os.environ.get('BDRT_RIDGE_START_ITER', 2))))
                    else:
                      extra.append(self._get_init_from_ridge(frequencies, Z, 'optimize', nonneg=nonneg, outliers=outliers,
                                                           inductance_scale=inductance_scale,
                                                           ridge_kw=dict({'max_iter': int(os.environ.get('BDRT_RIDGE_START_ITER', 2))}, **ridge_kw)))
            except Exception as e:                       # the ridge candidate is optional: the random start remains
                warnings.warn('ridge starting point not available (%s): MAP from the random start only' % e)
            finally:
                for k in self._RIDGE_SIDE_EFFECTS:
                    if k in saved:
                        setattr(self, k, saved[k])
                    elif hasattr(self, k):
                        delattr(self, k)
        want = (n_starts - 1) if n_starts is not None else (len(extra) if (single or init_from_ridge) else 3)
        draw = 1
        while len(extra) < want:
            extra.append(('random', draw)); draw += 1    # another draw of the random start
        return extra[:want]

    def _bayes_matrices(self, frequencies, Z, part, scale_Z, outliers, init_from_ridge, ridge_kw):
        """Matrices of the Bayesian fit ('discrete' penalty) and the resolved `outliers` flag ('auto': a ridge-based check
        decides, reference :1172-1187)."""
        def prep(f_, Z_):
            out = self._prep_matrices(f_, Z_, part, weights=None, dZ=False, scale_Z=scale_Z, penalty='discrete', fit_type='map')
            return out[0], out[1], out[6]
        frequencies, Z_scaled, dist_mat = prep(frequencies, Z)
        if outliers == 'auto':
            Z_sorted = self.Z_train
            idx = self.check_outliers(frequencies, Z_sorted, threshold=4, use_existing_fit=bool(init_from_ridge), **ridge_kw)
            outliers = len(idx) > 0
            if outliers:
                warnings.warn('Identified likely outliers at indices {}, f={} Hz. An outlier-robust error model will be '
                              'used. To disable this behavior, pass outliers=False.'.format(idx, frequencies[idx]))
            # check_outliers may have re-run ridge_fit with its own penalty: back to the matrices of the Bayesian fit
            frequencies, Z_scaled, dist_mat = prep(frequencies, Z_sorted)
        return frequencies, Z_scaled, dist_mat, outliers

    def _store_bayes_fit(self, model_type, mode, sigma_min, outliers):
        """Engine result -> the attributes the reference sets (:1223-1262): coefficients per distribution, offsets, error
        structure, fit type."""
        coef_key = {'Series': lambda info: 'x', 'Parallel': lambda info: 'x',
                    'Series-Parallel': lambda info: 'xs' if info['dist_type'] == 'series' else 'xp',
                    'Series-2Parallel': lambda info: 'xs' if info['dist_type'] == 'series' else 'xp%d' % info['order']}[model_type]
        self.distribution_fits = {}
        for name, info in self.distributions.items():
            if model_type in ('Series', 'Parallel') and info['dist_type'] != model_type.lower():
                continue
            self.distribution_fits[name] = {'coef': self._extract_parameter(coef_key(info), info['dist_type'], mode)}
            if model_type in ('Series', 'Parallel'):
                break
        self.R_inf = self._extract_parameter('Rinf', 'series', mode)
        self.inductance = self._extract_parameter('induc', 'series', mode)
        self.error_fit = {'sigma_min': self._rescale_coef(sigma_min, 'series')}
        for key, kind in (('sigma_tot', 'series'), ('sigma_res', 'series'), ('alpha_prop', None), ('alpha_re', None),
                          ('alpha_im', None)) + ((('sigma_out', 'series'),) if outliers == True else ()):
            self.error_fit[key] = self._extract_parameter(key, kind, mode)
        self.fit_type = 'map' if mode == 'optimize' else 'bayes'

    def drift_map_fit(self, *a, **k):
        raise NotImplementedError('drift fits: the reference ships no Stan model files for them (SURVEY section 2 row 15)')

    def _get_stan_model(self, nonneg, outliers, drift, drift_model, fitY, SA):
        """Model selection (reference :1566-1614)."""
        ns = len([1 for i in self.distributions.values() if i['dist_type'] == 'series'])
        npar = len([1 for i in self.distributions.values() if i['dist_type'] == 'parallel'])
        table = {(1, 0): 'Series', (0, 1): 'Parallel', (1, 1): 'Series-Parallel', (1, 2): 'Series-2Parallel'}
        if (ns, npar) not in table:
            raise NotImplementedError('The MultiDist model is a placeholder in the reference (no Stan file shipped)')
        s = table[(ns, npar)]
        if nonneg and ns >= 1:
            s += '_pos'
        if outliers:
            s += '_outliers'
        s += '_StanModel.pkl'
        return load_pickle(os.path.join(script_dir, 'stan_model_files', s)), s

    def _get_init_from_ridge(self, frequencies, Z, mode, nonneg, outliers, inductance_scale, ridge_kw):
        """Initial values from a deliberately under-fitted hyper-ridge solution (reference :1616-1682).  Returns the
        callable Stan's `init=` takes; it yields constrained values on the scaled-Z problem."""
        settings = dict(penalty='integral', hyper_lambda=True, lambda_0=1, hl_beta=5, weights='modulus')
        settings.update(ridge_kw)                       # user settings win
        self.ridge_fit(frequencies, Z, **settings)
        return self._init_values_from_ridge_fit(frequencies, Z, nonneg, outliers, inductance_scale)

    def _init_values_from_ridge_fit(self, frequencies, Z, nonneg, outliers, inductance_scale):
        """The second half of `_get_init_from_ridge`: Stan `init=` values from the ridge fit this instance holds."""
        name = next(iter(self.distributions))
        zs = self._Z_scale
        coef = self.distribution_fits[name]['coef']
        series = self.distributions[name]['dist_type'] == 'series'
        Rinf = self.R_inf / zs
        induc = max(self.inductance / zs, 0.0) or 1e-10  # lower=0 parameter: strictly positive start
        x0 = coef / zs if series else coef * zs
        if nonneg or not series:
            # lower=0 parameter: a coefficient the QP left at (numerically) zero starts a little inside the support instead
            # of making the initialisation fail (Stan rejects an initial value on the bound)
            x0 = np.maximum(x0, 1e-8 * max(float(np.max(x0)), 1e-300))
        values = {'x': x0, 'Rinf': Rinf, 'Rinf_raw': Rinf / 100, 'induc': induc,
                  'induc_raw': induc / inductance_scale}
        if outliers:
            suspects = self.check_outliers(frequencies, Z, threshold=3, use_existing_fit=True)   # lenient threshold
            if outliers is True or len(suspects) > 0:
                start = np.full(len(Z), 0.1)
                start[suspects] = 1
                values['sigma_out_raw'] = start
        return lambda: values

    def _prep_stan_data(self, frequencies, Z, part, model_type, dist_mat, outliers, sigma_min, mode, inductance_scale,
                        outlier_lambda, fitY, SA, SASY):
        """The Stan `data` dict with the calibrated hyper-parameter table (reference :1684-2122)."""
        if outlier_lambda is None:
            outlier_lambda = 10
        samp = mode == 'sample'
        ups_alpha, ups_beta = (1, 0.1) if samp else (0.05, 0.1)

        def Ls(m, kind):
            if samp:
                return m['L0'], m['L1'], 0.75 * m['L2']
            return (1.5 * (0.36 if kind == 'parallel_multi' else 0.24)) * m['L0'], 1.5 * 0.16 * m['L1'], 1.5 * 0.08 * m['L2']

        # part = 'real' / 'imag' (reference :1892-1905, :1973-...): the multi-distribution models keep N = 2 Nf and zero the
        # rows of the part that is not fitted, in A and in Z alike
        keep_re, keep_im = float(part != 'imag'), float(part != 'real')

        def stack(m):
            return np.concatenate((keep_re * m['A_re'], keep_im * m['A_im']))
        Z_stack = np.concatenate((keep_re * Z.real, keep_im * Z.imag))
        nf = len(frequencies)
        if part != 'both' and model_type in ['Series', 'Parallel']:
            # the reference hands Stan an Nf-row A and Z together with N = 2 Nf here (:1718-1723, :1739): pystan rejects that
            # data ("mismatch in dimension declared and found in context"), so there is no behaviour to reproduce
            raise ValueError("fit(part='%s') is only defined for multi-distribution models: for a single distribution the "
                             "reference's Stan data is dimensionally inconsistent and pystan rejects it" % part)
        common = {'N': 2 * nf, 'freq': frequencies, 'Z': Z_stack, 'N_tilde': 2 * nf, 'freq_tilde': frequencies,
                  'sigma_min': sigma_min, 'ups_alpha': ups_alpha, 'ups_beta': ups_beta, 'induc_scale': inductance_scale}
        if model_type in ['Series', 'Parallel']:
            name = [k for k, v in self.distributions.items() if v['dist_type'] == model_type.lower()][0]
            m = dist_mat[name]
            L0, L1, L2 = Ls(m, 'single')
            A = stack(m)
            dat = dict(common, K=A.shape[1], A=A, A_tilde=A, L0=L0, L1=L1, L2=L2)
            if outliers:
                dat.update(sigma_out_lambda=outlier_lambda, sigma_out_alpha=5 if samp else 2, sigma_out_beta=1)
        elif model_type == 'Series-Parallel':
            if len(self.distributions) > 2:
                raise ValueError('Too many distributions for Series-Parallel model')
            sn = [k for k, v in self.distributions.items() if v['dist_type'] == 'series'][0]
            pn = [k for k, v in self.distributions.items() if v['dist_type'] == 'parallel'][0]
            ms, mp = dist_mat[sn], dist_mat[pn]
            L0s, L1s, L2s = Ls(ms, 'single')
            L0p, L1p, L2p = Ls(mp, 'parallel_multi')
            As, Ap = stack(ms), stack(mp)
            dat = dict(common, Ks=As.shape[1], Kp=Ap.shape[1], As=As, Ap=Ap, As_tilde=As, Ap_tilde=Ap, L0s=L0s, L1s=L1s,
                       L2s=L2s, L0p=L0p, L1p=L1p, L2p=L2p, x_sum_invscale=1 if samp else 0.,
                       xp_scale=self.distributions[pn].get('x_scale', 1))
            if outliers:
                dat['so_invscale'] = outlier_lambda
        elif model_type == 'Series-2Parallel':
            sn = [k for k, v in self.distributions.items() if v['dist_type'] == 'series'][0]
            p1, p2 = sorted([k for k, v in self.distributions.items() if v['dist_type'] == 'parallel'])
            self.distributions[p1]['order'] = 1
            self.distributions[p2]['order'] = 2
            ms, m1, m2 = dist_mat[sn], dist_mat[p1], dist_mat[p2]
            L0s, L1s, L2s = Ls(ms, 'single')
            a = Ls(m1, 'parallel_multi'); b = Ls(m2, 'parallel_multi')
            As, A1, A2 = stack(ms), stack(m1), stack(m2)
            dat = dict(common, Ks=As.shape[1], Kp1=A1.shape[1], Kp2=A2.shape[1], As=As, Ap1=A1, Ap2=A2, As_tilde=As,
                       Ap1_tilde=A1, Ap2_tilde=A2, L0s=L0s, L1s=L1s, L2s=L2s, L0p1=a[0], L1p1=a[1], L2p1=a[2],
                       L0p2=b[0], L1p2=b[1], L2p2=b[2], x_sum_invscale=0.1 if samp else 0.,
                       xp1_scale=self.distributions[p1].get('x_scale', 1), xp2_scale=self.distributions[p2].get('x_scale', 1))
            if outliers:
                dat['so_invscale'] = outlier_lambda
        else:
            raise NotImplementedError('MultiDist is a placeholder in the reference')
        return dat

    # ================================================================== matrices / scaling (reference :2127-2450)
    def _prep_matrices(self, frequencies, Z, part, weights, dZ, scale_Z, penalty, fit_type, sort_desc=True):
        if len(frequencies) != len(Z):
            raise ValueError("Length of frequencies and Z must be equal")
        Z = np.array(Z) if type(Z) != np.ndarray else Z
        frequencies = np.array(frequencies, dtype=float) if type(frequencies) != np.ndarray else frequencies
        _validate_spectrum(frequencies, Z)
        if sort_desc:
            order = np.argsort(frequencies)[::-1]
            frequencies, Z = frequencies[order], Z[order]
        self.Z_train = Z
        if not _same(self.distributions, self._cached_distributions):
            self._recalc_mat = True
            self.f_pred = None
        freq_subset = False
        ft = np.asarray(self.f_train, dtype=float)
        same_grid = len(ft) == len(frequencies) and bool(np.min(rel_round(ft, 10) == rel_round(frequencies, 10)))
        if not same_grid:
            rt = set(rel_round(ft, 10).tolist()) if len(ft) > 1 else set()
            if len(rt) and all(v in rt for v in rel_round(frequencies, 10).tolist()):
                freq_subset = True
            else:
                self.f_train = frequencies
                self._recalc_mat = True
        else:
            self.f_train = frequencies
        if scale_Z:
            Z = self._scale_Z(Z, fit_type)
            if type(weights) in (list, np.ndarray):
                weights = np.array(weights) / self._Z_scale
        else:
            self._Z_scale = 1
        weights = self._format_weights(frequencies, Z, weights, part)
        W_re, W_im = np.diag(np.real(weights)), np.diag(np.imag(weights))
        dist_mat = {}
        for name, info in self.distributions.items():
            tmp = deepcopy(self.distributions)
            bf = info.get('basis_freq', self.basis_freq)
            if bf is None:
                # one decade beyond the measured range on each side, 10 points per decade (:2191-2197); the
                # int() truncation is reproduced with the same float64 operation order
                tmin = np.log10(1 / (2 * np.pi * np.max(frequencies))) - 1
                tmax = np.log10(1 / (2 * np.pi * np.min(frequencies))) + 1
                num_decades = tmax - tmin
                tau = np.logspace(tmin, tmax, int(10 * num_decades + 1))
            else:
                tau = 1 / (2 * np.pi * np.asarray(bf, dtype=float))
            tmp[name]['tau'] = tau
            if info.get('epsilon', self.epsilon) is None:
                tmp[name]['epsilon'] = 1 / np.mean(np.diff(np.log(tau)))
            elif info.get('epsilon', None) is None:
                tmp[name]['epsilon'] = self.epsilon
            epsilon = tmp[name].get('epsilon', self.epsilon)
            keep = self._recalc_mat
            self.distributions = tmp
            self._recalc_mat = keep
            info = self.distributions[name]
            akw = dict(tau=tau, basis=self.basis, epsilon=epsilon, kernel=info['kernel'], dist_type=info['dist_type'],
                       symmetry=info.get('symmetry', ''), bc=info.get('bc', ''), ct=info.get('ct', False),
                       k_ct=info.get('k_ct', None))
            store = self.distribution_matrices.setdefault(name, {})
            if self._recalc_mat or 'A_re' not in store or 'A_im' not in store:
                store['A_re'] = construct_A(frequencies, 'real', fit_inductance=self.fit_inductance, **akw)
                store['A_im'] = construct_A(frequencies, 'imag', fit_inductance=self.fit_inductance, **akw)
                store.pop('B', None)
                A_re, A_im = store['A_re'].copy(), store['A_im'].copy()
            elif freq_subset:
                rt = rel_round(np.asarray(self.f_train, dtype=float), 10)
                idx = np.array([np.where(rt == v)[0][0] for v in rel_round(frequencies, 10)])
                A_re, A_im = store['A_re'][idx, :].copy(), store['A_im'][idx, :].copy()
            else:
                A_re, A_im = store['A_re'].copy(), store['A_im'].copy()
            B = None
            if dZ and info['kernel'] == 'DRT':
                if 'B' not in store:
                    # first difference of A' on a half-shifted tau grid: B@coef ~ dZ'/dln(tau) (:2273-2285)
                    step = np.mean(np.diff(np.log(tau)))
                    edges = np.logspace(np.log10(np.exp(np.log(tau[0]) - step / 2)), np.log10(np.exp(np.log(tau[-1]) + step / 2)),
                                        len(tau) + 1)
                    Bp = construct_A(1 / (2 * np.pi * edges), 'real', **akw)
                    store['B'] = Bp[1:, :] - Bp[:-1, :]
                B = store['B'].copy()
            dist_mat[name] = {}
            fb = 1 / (2 * np.pi * tau)
            # (the penalty matrices depend on the basis grid only: kept with the A matrices while the grid stands -- a batch of
            # spectra on one grid, `fit_many`, builds them once)
            pen_key = (self.basis, float(epsilon), len(tau), float(tau[0]), float(tau[-1]))
            pen_names = {'integral': ('M0', 'M1', 'M2'), 'discrete': ('L0', 'L1', 'L2'),
                         'cholesky': ('M0', 'M1', 'M2', 'L0', 'L1', 'L2')}
            if penalty not in pen_names:
                raise ValueError(f'Invalid penalty argument {penalty}. Options are integral, discrete, and cholesky')
            if self._recalc_mat:
                store.pop('_penalty', None)
            # one cache entry per penalty kind (a MAP batch alternates 'integral' -- its ridge start -- and 'discrete'); the
            # arrays are handed out as copies, like A_re / A_im, so that nothing downstream can edit the store of all views
            cache = store.setdefault('_penalty', {})
            hit = cache.get(penalty)
            if hit is not None and hit[0] == pen_key:
                mats = hit[1]
            else:
                mats = {}
                if penalty == 'integral':
                    for o in (0, 1, 2):
                        mats['M%d' % o] = construct_M(fb, basis=self.basis, order=o, epsilon=epsilon)
                elif penalty == 'discrete':
                    for o in (0, 1, 2):
                        mats['L%d' % o] = construct_L(fb, tau=tau, basis=self.basis, epsilon=epsilon, order=o)
                else:
                    # M = L^T L with L upper triangular, so that x^T M x = ||L x||^2 (reference :2309-2317)
                    from scipy.linalg import cholesky
                    for o in (0, 1, 2):
                        M = construct_M(fb, basis=self.basis, order=o, epsilon=epsilon)
                        mats['M%d' % o] = M
                        mats['L%d' % o] = cholesky(M)
                cache[penalty] = (pen_key, mats)
            for k_ in pen_names[penalty]:
                dist_mat[name][k_] = mats[k_].copy()
            store.update(mats)
            dist_mat[name].update({'A_re': A_re, 'A_im': A_im, 'WA_re': W_re @ A_re, 'WA_im': W_im @ A_im, 'B': B})
        self._recalc_mat = False
        self._cached_distributions = self.distributions.copy()
        return frequencies, Z, W_re @ Z.real, W_im @ Z.imag, W_re, W_im, dist_mat

    # named weighting schemes -> complex weight vector (real part weights Z', imaginary part Z''); reference :2351-2366
    _WEIGHT_SCHEMES = {
        'unity': lambda Z: np.ones(len(Z)) * (1 + 1j),
        'modulus': lambda Z: (1 + 1j) / np.sqrt(np.real(Z * Z.conjugate())),
        'Orazem': lambda Z: (1 + 1j) / (np.abs(Z.real) + np.abs(Z.imag)),
        'proportional': lambda Z: 1 / np.abs(Z.real) + 1j / np.abs(Z.imag),
        'prop_adj': lambda Z: (lambda floor: 1 / (np.abs(Z.real) + floor) + 1j / (np.abs(Z.imag) + floor))(
            np.percentile(np.real(Z * Z.conjugate()), 25)),
    }

    def _format_weights(self, frequencies, Z, weights, part):
        """Complex weight vector: real part weights Z', imaginary part Z'' (reference :2338-2395).
        weights: None | scheme name | real or complex scalar | array (real: both parts; complex: one part each)."""
        if part not in ('both', 'real', 'imag'):
            raise ValueError(f"Invalid part {part}. Options are 'both', 'real', or 'imag'")
        n = len(frequencies)
        if weights is None:
            weights = 'unity'
        if isinstance(weights, str):
            scheme = self._WEIGHT_SCHEMES.get(weights)
            if scheme is None:
                raise ValueError(f"Invalid weights argument {weights}. String options are 'unity', 'modulus', "
                                 f"'proportional', and 'prop_adj'")
            w = scheme(Z)
        elif type(weights) in (float, int):
            w = np.full(n, weights * (1 + 1j))
        elif type(weights) == complex:
            w = np.full(n, weights)
        else:
            if len(weights) != n:
                raise ValueError("Weights array must match length of data")
            w = weights
        all_real = bool(np.min(np.isreal(w)))
        one = np.ones(n)
        if part == 'real':                       # the imaginary part is not fitted: unit weight
            return np.real(w) + 1j * one
        if not all_real:                         # complex weights already carry one weight per part
            return w
        return w + 1j * w if part == 'both' else one + 1j * w

    def _scale_Z(self, Z, fit_type):
        """_Z_scale = std|Z| / sqrt(N/81); pure parallel planar DDT scales the admittance instead (reference :2411-2443)."""
        ns = len([1 for i in self.distributions.values() if i['dist_type'] == 'series'])
        npar = len([1 for i in self.distributions.values() if i['dist_type'] == 'parallel'])
        Zmod = np.abs(Z)
        self._Z_scale = np.std(Zmod) / np.sqrt(len(Z) / 81)
        if npar == 1 and ns == 0 and fit_type != 'ridge':
            info = [i for i in self.distributions.values() if i['dist_type'] == 'parallel'][0]
            if info['kernel'] == 'DDT' and info['symmetry'] == 'planar':
                target = 14 if info['bc'] == 'transmissive' else 2.4
                self._Z_scale = target * np.sqrt(len(Z) / 81) / np.std(np.abs(1 / Z))
        if not (np.isfinite(self._Z_scale) and self._Z_scale > 0):
            raise ValueError('the modulus of Z has no spread (std |Z| = %r): the spectrum cannot be scaled; pass scale_Z=False '
                             'or check the data' % (self._Z_scale,))
        return Z / self._Z_scale

    def _rescale_coef(self, coef, dist_type):
        if dist_type == 'series':
            return coef * self._Z_scale
        elif dist_type == 'parallel':
            return coef / self._Z_scale

    # ================================================================== result extraction (reference :2494-2566)
    def _extract_parameter(self, stan_key, dist_type, mode):
        unscaled = stan_key in ['alpha_prop', 'alpha_re', 'alpha_im']
        if mode == 'optimize':
            v = self._opt_result[stan_key]
            return v if unscaled else self._rescale_coef(v, dist_type)
        v = self._sample_result[stan_key]
        return np.mean(v) if unscaled else self._rescale_coef(np.mean(v, axis=0), dist_type)

    def _get_stan_coef_name(self, distribution_name):
        dist_type = self.distributions[distribution_name]['dist_type']
        model_type = self.stan_model_name.split('_')[0]
        if model_type in ['Series', 'Parallel']:
            return 'x'
        if dist_type == 'series':
            return 'xs'
        if model_type == 'Series-Parallel':
            return 'xp'
        return 'xp%d' % self.distributions[distribution_name]['order']

    def coef_percentile(self, distribution_name, percentile):
        """Percentile of each coefficient over the draws (then rescaled) -- per coefficient, as the reference does (:2547-2566, H6)."""
        if self.fit_type != 'bayes':
            raise ValueError('Percentile prediction is only available for bayes_fit')
        coef = post.percentile(self._sample_result[self._get_stan_coef_name(distribution_name)], percentile, axis=0)
        return self._rescale_coef(coef, self.distributions[distribution_name]['dist_type'])

    # ================================================================== prediction (reference :2571-3311)
    def _get_prediction_matrices(self, frequencies, distributions):
        frequencies = np.asarray(frequencies, dtype=float)
        rf = rel_round(frequencies, 10)

        def lookup(src_f, src):
            rs = rel_round(np.asarray(src_f, dtype=float), 10)
            if len(rs) == len(rf) and bool(np.min(rs == rf)):
                return {n: {'A_re': src[n]['A_re'].copy(), 'A_im': src[n]['A_im'].copy()} for n in distributions}
            pos = {v: i for i, v in enumerate(rs.tolist())}
            if all(v in pos for v in rf.tolist()):
                idx = np.array([pos[v] for v in rf.tolist()])
                return {n: {'A_re': src[n]['A_re'][idx, :].copy(), 'A_im': src[n]['A_im'][idx, :].copy()} for n in distributions}
            return None

        def build():
            out = {}
            for n in distributions:
                info = self.distributions[n]
                kw = dict(tau=info['tau'], basis=self.basis, fit_inductance=self.fit_inductance, epsilon=info['epsilon'],
                          kernel=info['kernel'], dist_type=info['dist_type'], symmetry=info.get('symmetry', ''),
                          bc=info.get('bc', ''), ct=info.get('ct', False), k_ct=info.get('k_ct', None))
                out[n] = {'A_re': construct_A(frequencies, 'real', **kw), 'A_im': construct_A(frequencies, 'imag', **kw)}
            return out
        if self.f_pred is not None and all(n in getattr(self, 'prediction_matrices', {}) for n in distributions):
            pm = lookup(self.f_pred, self.prediction_matrices)
            if pm is None:
                pm = build()
                self.prediction_matrices, self.f_pred = pm, frequencies
            return pm
        have = all(len(self.distribution_matrices.get(n, {})) > 0 and 'A_re' in self.distribution_matrices[n] for n in distributions)
        pm = lookup(self.f_train, self.distribution_matrices) if have and len(np.atleast_1d(self.f_train)) > 1 else None
        if pm is None:
            pm = build()
        self.f_pred, self.prediction_matrices = frequencies, pm
        return pm

    def _dist_list(self, distributions):
        if distributions is None:
            return [k for k in self.distribution_fits.keys()]
        return [distributions] if type(distributions) == str else list(distributions)

    def predict_Z_distribution(self, frequencies, distributions=None, include_offsets=True):
        """Impedance of every posterior draw (rows) (reference :2963-3031)."""
        if self.fit_type != 'bayes':
            raise ValueError('predict_Z_distribution is only available for bayes_fit results')
        frequencies = np.asarray(frequencies, dtype=float)
        distributions = self._dist_list(distributions)
        full = len(distributions) == len(self.distributions) and include_offsets
        if not full:
            warnings.warn('All distributions and offsets should be included for meaningful results from predict_Z_distribution')
        ft = np.asarray(self.f_train, dtype=float)
        if full and len(ft) == len(frequencies) and bool(np.min(rel_round(ft, 10) == rel_round(frequencies, 10))):
            Zs = self._sample_result['Z_hat'] * self._Z_scale
            return Zs[:, :len(frequencies)] + 1j * Zs[:, len(frequencies):]
        pm = self._get_prediction_matrices(frequencies, distributions)
        ns = len(self._sample_result['Rinf'])
        Zm = np.zeros((ns, len(frequencies)), dtype=complex)
        for name, mat in pm.items():
            dt = self.distributions[name]['dist_type']
            cm = self._rescale_coef(self._sample_result[self._get_stan_coef_name(name)], dt)
            v = cm @ mat['A_re'].T + 1j * (cm @ mat['A_im'].T)
            Zm += v if dt == 'series' else 1 / v
        if include_offsets:
            Zm += self._rescale_coef(self._sample_result['Rinf'], 'series')[:, None]
            Zm += 1j * 2 * np.pi * frequencies * self._rescale_coef(self._sample_result['induc'], 'series')[:, None]
        return Zm

    def predict_Z(self, frequencies, times=None, distributions=None, include_offsets=True, percentile=None):
        """Impedance predicted by the fitted distributions (reference :2669-2961)."""
        frequencies = np.asarray(frequencies, dtype=float)
        if times is not None:
            raise NotImplementedError('drift predictions are out of scope')
        distributions = self._dist_list(distributions)
        if percentile is not None:
            if self.fit_type != 'bayes':
                raise ValueError('Percentile prediction is only available for bayes_fit results')
            full = len(distributions) == len(self.distributions) and include_offsets
            if not full:
                warnings.warn('If percentile is specified, all distributions and offsets should be included for meaningful results')
            ft = np.asarray(self.f_train, dtype=float)
            if full and len(ft) == len(frequencies) and bool(np.min(rel_round(ft, 10) == rel_round(frequencies, 10))):
                Zp = post.percentile(self._sample_result['Z_hat'], percentile, axis=0) * self._Z_scale
                return Zp[:len(frequencies)] + 1j * Zp[len(frequencies):]
            if full and len(distributions) == 1 and self.distributions[distributions[0]]['dist_type'] == 'series':
                # one series distribution: Z of a draw is affine in (Rinf, induc, coef): project and reduce on the GPU
                name = distributions[0]
                mat = self._get_prediction_matrices(frequencies, distributions)[name]
                cm = self._rescale_coef(self._sample_result[self._get_stan_coef_name(name)], 'series')
                X = np.column_stack([self._rescale_coef(self._sample_result['Rinf'], 'series'),
                                     self._rescale_coef(self._sample_result['induc'], 'series'), cm])
                nfq = len(frequencies)
                Phi = np.zeros((2 * nfq, X.shape[1]))
                Phi[:nfq, 0] = 1.0
                Phi[nfq:, 1] = 2 * np.pi * frequencies
                Phi[:nfq, 2:] = mat['A_re']
                Phi[nfq:, 2:] = mat['A_im']
                Zq = post.project_percentile(X, Phi, None, percentile)
                return Zq[..., :nfq] + 1j * Zq[..., nfq:]
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                Zm = self.predict_Z_distribution(frequencies, distributions, include_offsets)
            return post.percentile(Zm.real, percentile, axis=0) + 1j * post.percentile(Zm.imag, percentile, axis=0)
        pm = self._get_prediction_matrices(frequencies, distributions)
        Zp = np.zeros(len(frequencies), dtype=complex)
        for name, mat in pm.items():
            coef = self.distribution_fits[name]['coef']
            v = mat['A_re'] @ coef + 1j * (mat['A_im'] @ coef)
            Zp += v if self.distributions[name]['dist_type'] == 'series' else 1 / v
        if include_offsets:
            Zp += self.R_inf
            Zp += 1j * 2 * np.pi * frequencies * self.inductance
        return Zp

    def predict_Rp(self, distributions=None, percentile=None, time=None):
        """Polarisation resistance (reference :3033-3087)."""
        distributions = self._dist_list(distributions)
        if len(distributions) > 1:
            ends = np.array([1e20, 1e-20])
            if percentile is None:
                Zr = self.predict_Z(ends, distributions=distributions)
                return np.real(Zr[1] - Zr[0])
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                Zm = self.predict_Z_distribution(ends, distributions=distributions)
            return post.percentile(np.real(Zm[:, 1] - Zm[:, 0]), percentile)
        info = self.distributions[distributions[0]]
        if info['kernel'] == 'DRT' and 'coef' in self.distribution_fits[distributions[0]]:
            if percentile is None:                         # area under the DRT
                return np.sum(self.distribution_fits[distributions[0]]['coef']) * np.pi ** 0.5 / info['epsilon']
            if self.fit_type != 'bayes':
                raise ValueError('Percentile prediction is only available for bayes_fit results')
            cm = self._rescale_coef(self._sample_result[self._get_stan_coef_name(distributions[0])], 'series')
            return post.percentile(np.sum(cm, axis=1) * np.pi ** 0.5 / info['epsilon'], percentile)
        ends = np.array([1e20, 1e-20])
        if percentile is None:
            Zr = self.predict_Z(ends, distributions=distributions)
            return np.real(Zr[1] - Zr[0])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Zm = self.predict_Z_distribution(ends, distributions=distributions)
        return post.percentile(np.real(Zm[:, 1] - Zm[:, 0]), percentile)

    def predict_sigma(self, frequencies, percentile=None, times=None):
        """Error scale (sigma_re, sigma_im) of the fitted error model (reference :3089-3139)."""
        frequencies = np.asarray(frequencies, dtype=float)
        if percentile is not None and self.fit_type != 'bayes':
            raise ValueError('Percentile prediction is only available for bayes_fit')
        if self.fit_type not in ('bayes', 'map'):
            raise ValueError('Error scale prediction only available for bayes_fit and map_fit')
        ft = np.asarray(self.f_train, dtype=float)
        nf = len(ft)
        if nf == len(frequencies) and bool(np.min(rel_round(ft, 10) == rel_round(frequencies, 10))):
            if self.fit_type == 'bayes' and percentile is not None:
                st = post.percentile(self._sample_result['sigma_tot'], percentile, axis=0) * self._Z_scale
            else:
                st = self.error_fit['sigma_tot']
            return st[:nf].copy(), st[nf:].copy()
        if self.fit_type == 'bayes' and percentile is not None:
            sres = post.percentile(self._sample_result['sigma_res'], percentile) * self._Z_scale
            ap, ar, ai = [post.percentile(self._sample_result[k], percentile) for k in ('alpha_prop', 'alpha_re', 'alpha_im')]
            try:
                sout = post.percentile(self._sample_result['sigma_out'], percentile, axis=0) * self._Z_scale
            except (ValueError, KeyError):
                sout = np.zeros(2 * nf)
        else:
            sres, ap, ar, ai = [self.error_fit[k] for k in ('sigma_res', 'alpha_prop', 'alpha_re', 'alpha_im')]
            sout = self.error_fit.get('sigma_out', np.zeros(2 * nf))
        Zp = self.predict_Z(frequencies, percentile=percentile)
        base2 = sres ** 2 + np.min(sout) ** 2 + self.error_fit['sigma_min'] ** 2
        common = (ar * Zp.real) ** 2 + (ai * Zp.imag) ** 2
        return np.sqrt(base2 + (ap * Zp.real) ** 2 + common), np.sqrt(base2 + (ap * Zp.imag) ** 2 + common)

    def predict_distribution(self, name=None, eval_tau=None, percentile=None, time=None):
        """gamma(tau) = sum_m coef_m exp(-(eps ln(tau/tau_m))^2) (reference :3162, :3298-3311)."""
        if time is not None:
            raise NotImplementedError('drift predictions are out of scope')
        if name is None:
            name = list(self.distributions.keys())[0]
        info = self.distributions[name]
        if eval_tau is None:
            eval_tau = info['tau']
        coef = self.coef_percentile(name, percentile) if percentile is not None else self.distribution_fits[name]['coef']
        y = np.log(np.asarray(eval_tau, dtype=float)[:, None] / info['tau'][None, :])
        return _gaussian(y, info['epsilon']) @ coef

    def score(self, frequencies, Z, metric='chi_sq', weights=None, part='both', times=None):
        """Fit quality: weighted chi-square or r2 of the predicted impedance (reference :3141-3160)."""
        frequencies, Z = np.asarray(frequencies, dtype=float), np.asarray(Z)
        w = self._format_weights(frequencies, Z, weights, part)
        Zp = self.predict_Z(frequencies)
        if part == 'both':
            y = np.concatenate((Z.real, Z.imag)); yh = np.concatenate((Zp.real, Zp.imag)); ww = np.concatenate((w.real, w.imag))
        else:
            y, yh, ww = getattr(Z, part), getattr(Zp, part), getattr(w, part)
        if metric == 'chi_sq':
            return np.sum(((yh - y) * ww) ** 2) / len(frequencies)
        if metric == 'r2':
            ss_res = np.sum(ww ** 2 * (yh - y) ** 2)
            ss_tot = np.sum(ww ** 2 * (y - np.average(y, weights=ww ** 2)) ** 2)
            return 1 - ss_res / ss_tot
        raise ValueError(f"Invalid metric {metric}. Options are 'chi_sq', 'r2'")

    # ================================================================== outliers (reference :3313-3376)
    def check_outliers(self, frequencies, Z, threshold, use_existing_fit, **ridge_kw):
        """Indices of likely outliers: IQR test on ridge residuals, or z-score with the fitted error model."""
        frequencies, Z = np.asarray(frequencies), np.asarray(Z)
        fit_exists = (_same(frequencies, self.f_train) and _same(Z, self.Z_train) and not self._recalc_mat
                      and hasattr(self, 'distribution_fits') and len(self.distribution_fits) > 0)
        if not (use_existing_fit and fit_exists):
            self.ridge_fit(frequencies, Z, preset='Huang', **ridge_kw)
            order = np.argsort(frequencies)[::-1]
            frequencies, Z = frequencies[order], Z[order]
        Z_err = self.predict_Z(frequencies) - Z
        if self.fit_type == 'ridge':
            Zmod = np.sqrt(Z.real ** 2 + Z.imag ** 2)
            re_t = get_outlier_thresh(np.abs(Z_err.real / Zmod), iqr_factor=threshold)
            im_t = get_outlier_thresh(np.abs(Z_err.imag / Zmod), iqr_factor=threshold)
            return np.argwhere((Z_err.real / Zmod) ** 2 + (Z_err.imag / Zmod) ** 2 >= re_t ** 2 + im_t ** 2)
        s_re, s_im = self.predict_sigma(frequencies)
        zs = np.sqrt(((Z_err.real / s_re) ** 2 + (Z_err.imag / s_im) ** 2) / 2)
        return np.argwhere(zs > threshold)

    # ================================================================== persistence (reference :3980-4064), arrays only
    def get_fit_attributes(self, which='all'):
        """Names of the attributes that make up a stored fit (same sets as the reference, :3980-4002)."""
        fit_attributes = {
            'common': {'core': ['distributions', 'distribution_fits', 'f_train', 'Z_train', '_Z_scale', 'fit_type', 'R_inf',
                                'inductance'],
                       'detail': ['distribution_matrices']},
            'ridge': {'core': [], 'detail': ['_iter_history']},
            'map': {'core': ['stan_model_name', 'error_fit'], 'detail': ['_stan_input', '_init_params', '_opt_result']},
            'bayes': {'core': ['stan_model_name', '_sample_result', 'error_fit'], 'detail': ['_stan_input', '_init_params']},
        }
        if which == 'all':
            return sum(fit_attributes['common'].values(), []) + sum(fit_attributes[self.fit_type].values(), [])
        if which not in ('core', 'detail'):
            raise ValueError("which must be 'core', 'detail' or 'all'")
        return fit_attributes['common'][which] + fit_attributes[self.fit_type][which]

    def get_fit_data(self, which='all'):
        out = {}
        for k in self.get_fit_attributes(which):
            if not hasattr(self, k):
                continue                                   # e.g. _init_params of a fit without ridge initialisation
            v = getattr(self, k)
            if k == '_sample_result':
                v = v.to_saved()                           # plain arrays instead of a live GPU-backed fit object
            elif k == '_iter_history':
                v = [{kk: vv for kk, vv in h.items() if kk != 'result'} | {'fun': h.get('fun')} for h in v]
            out[k] = deepcopy(v)
        return out

    def save_fit_data(self, filename=None, which='all'):
        """Store the fit (reference :4004-4036): a pickled dict of plain numpy containers; returned when filename is None."""
        import pickle
        data = self.get_fit_data(which)
        if filename is None:
            return data
        with open(filename, 'wb') as f:
            pickle.dump(data, f, pickle.DEFAULT_PROTOCOL)

    def load_fit_data(self, data):
        """Restore a stored fit (reference :4038-4064): file name or dict, as produced by `save_fit_data`."""
        import pickle
        if isinstance(data, str):
            with open(data, 'rb') as f:
                data = pickle.load(f)
        f_pred_old = deepcopy(self.f_pred)
        self._cached_distributions = self.distributions.copy()
        for k, v in dict(data).items():
            if k == '_sample_result' and not hasattr(v, 'chain_draws'):
                from .engine import SavedFit
                v = SavedFit(v)
            setattr(self, k, v)
        if 'distribution_matrices' not in data:
            # the stored fit carries no matrices: prediction matrices can be kept only if the distributions are the same
            same = set(self.distributions) == set(self._cached_distributions) and all(
                _dist_equal(self.distributions[n], self._cached_distributions[n]) for n in self.distributions)
            self.f_pred = f_pred_old if same else None

