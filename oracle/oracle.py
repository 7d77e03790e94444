"""ctypes front-end of the CPU ORACLE (oracle/bdrt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (bayes_drt_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAXB = 3

KERNEL_IDS = {
    ('DRT', None, None): 0,
    ('DDT', 'blocking', 'planar'): 1,
    ('DDT', 'blocking', 'spherical'): 2,
    ('DDT', 'transmissive', 'planar'): 3,
}


class _Model(C.Structure):
    _fields_ = [('nf', C.c_int), ('nblocks', C.c_int),
                ('K', C.c_int * MAXB), ('is_parallel', C.c_int * MAXB), ('nonneg', C.c_int * MAXB),
                ('x_scale', C.c_double * MAXB),
                ('A', C.c_void_p * MAXB), ('L0', C.c_void_p * MAXB), ('L1', C.c_void_p * MAXB),
                ('L2', C.c_void_p * MAXB),
                ('Z', C.c_void_p), ('freq', C.c_void_p),
                ('sigma_min', C.c_double), ('ups_alpha', C.c_double), ('ups_beta', C.c_double),
                ('induc_scale', C.c_double),
                ('outlier_mode', C.c_int),
                ('so_lambda', C.c_double), ('so_alpha', C.c_double), ('so_beta', C.c_double),
                ('use_x_sum', C.c_int), ('x_sum_invscale', C.c_double)]


def build(force=False, native=False):
    """liboracle.so (the checker, -O2, host independent) or, native=True, liboracle_native.so (-O3 -march=native: the CPU
    baseline of bench.py; always rebuilt with force=True there because -march=native code does not travel)."""
    name = 'liboracle_native.so' if native else 'liboracle.so'
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ('bdrt_oracle.c', 'bdrt_oracle.h', 'nuts_oracle.c')]
    if force and os.path.exists(so):
        os.remove(so)
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-s', '-C', _HERE, name])
    return so


def lib():
    global _LIB
    if _LIB is None:
        # BDRT_ORACLE_NATIVE=1 (set by bench.py's cpu_baseline workers only): the -O3 -march=native build
        _LIB = C.CDLL(build(native=bool(os.environ.get('BDRT_ORACLE_NATIVE'))))
        _LIB.orc_num_params.restype = C.c_int
        _LIB.orc_logp_grad.restype = C.c_int
        _LIB.orc_forward.restype = C.c_int
        _LIB.orc_build_A.restype = C.c_int
        _LIB.orc_build_A_basis.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class OracleModel:
    """A Stan model instance = blocks + data.  blocks: list of dicts with keys
    A [2nf x K], L0, L1, L2 [K x K] (already mode-scaled), parallel (bool), nonneg (bool), x_scale."""

    def __init__(self, blocks, Z, freq, sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0,
                 outlier_mode=0, so_lambda=10.0, so_alpha=5.0, so_beta=1.0, use_x_sum=None, x_sum_invscale=0.0):
        self._keep = []
        m = _Model()
        m.nf = len(freq)
        m.nblocks = len(blocks)
        for b, blk in enumerate(blocks):
            A = _f64(blk['A'])
            assert A.shape[0] == 2 * m.nf
            m.K[b] = A.shape[1]
            m.is_parallel[b] = int(bool(blk.get('parallel', False)))
            m.nonneg[b] = int(bool(blk.get('nonneg', False)))
            m.x_scale[b] = float(blk.get('x_scale', 1.0))
            arrs = [A] + [_f64(blk[k]) for k in ('L0', 'L1', 'L2')]
            self._keep += arrs
            m.A[b], m.L0[b], m.L1[b], m.L2[b] = [a.ctypes.data for a in arrs]
        self.Z = _f64(Z)
        self.freq = _f64(freq)
        m.Z = self.Z.ctypes.data
        m.freq = self.freq.ctypes.data
        m.sigma_min, m.ups_alpha, m.ups_beta, m.induc_scale = sigma_min, ups_alpha, ups_beta, induc_scale
        m.outlier_mode = outlier_mode
        m.so_lambda, m.so_alpha, m.so_beta = so_lambda, so_alpha, so_beta
        if use_x_sum is None:
            use_x_sum = len(blocks) > 1
        m.use_x_sum = int(use_x_sum)
        m.x_sum_invscale = x_sum_invscale
        self.m = m
        self.D = lib().orc_num_params(C.byref(m))
        self.Ks = [m.K[b] for b in range(m.nblocks)]

    def layout(self):
        o_x = (C.c_int * MAXB)(); o_u = (C.c_int * MAXB)(); o_d = (C.c_int * MAXB)()
        o_err = C.c_int(); o_so = C.c_int()
        pos = np.zeros(self.D, dtype=np.uint8)
        lib().orc_layout(C.byref(self.m), o_x, C.byref(o_err), C.byref(o_so), o_u, o_d, _p(pos))
        nb = self.m.nblocks
        return dict(x=list(o_x)[:nb], err=o_err.value, so=o_so.value, ups=list(o_u)[:nb], d=list(o_d)[:nb],
                    is_pos=pos.astype(bool))

    def logp_grad(self, theta, jacobian=True):
        theta = _f64(theta)
        lp = C.c_double()
        g = np.empty(self.D)
        lib().orc_logp_grad(C.byref(self.m), _p(theta), int(jacobian), C.byref(lp), _p(g))
        return lp.value, g

    def logp(self, theta, jacobian=True):
        theta = _f64(theta)
        lp = C.c_double()
        lib().orc_logp_grad(C.byref(self.m), _p(theta), int(jacobian), C.byref(lp), None)
        return lp.value

    def forward(self, theta):
        theta = _f64(theta)
        nf = self.m.nf
        Kt = sum(self.Ks)
        out = dict(Z_hat=np.empty(2 * nf), sigma_tot=np.empty(2 * nf), q=np.empty(Kt), ups=np.empty(Kt),
                   dups=np.empty(Kt - 2 * len(self.Ks)))
        xs = C.c_double()
        lib().orc_forward(C.byref(self.m), _p(theta), _p(out['Z_hat']), _p(out['sigma_tot']), _p(out['q']),
                          _p(out['ups']), _p(out['dups']), C.byref(xs))
        out['x_sum'] = xs.value
        return out

    def constrain(self, theta):
        out = np.empty(self.D)
        lib().orc_constrain(C.byref(self.m), _p(_f64(theta)), _p(out))
        return out

    def unconstrain(self, params):
        out = np.empty(self.D)
        lib().orc_unconstrain(C.byref(self.m), _p(_f64(params)), _p(out))
        return out


# ------------------------------------------------------------------------------------------ matrices
def _rel_round(x, precision):
    """bayes_drt/utils.py:113-130 (relative rounding used for frequency matching)."""
    x = np.asarray(x, dtype=float)
    scale = np.floor(np.log10(x + 1e-30))
    digits = (precision - scale).astype(int)
    return np.array([round(float(v), int(d)) for v, d in zip(x, digits)])


def is_loguniform(frequencies):
    """bayes_drt/utils.py:133-139."""
    fd = np.diff(np.log(frequencies))
    return bool(np.std(fd) / np.mean(fd) <= 0.01)


def a_is_toeplitz(frequencies, tau, ct=False):
    """Toeplitz decision of construct_A (bayes_drt/matrices.py:145-205)."""
    omega = np.asarray(frequencies, dtype=float) * 2 * np.pi
    tau = np.asarray(tau, dtype=float)
    inv_om = _rel_round(1 / omega, 10)
    tau_r = _rel_round(tau, 10)
    tau_eq_omega = len(tau) == len(omega) and bool(np.all(tau_r == inv_om))
    subset = False
    hit = np.where(tau_r == inv_om[0])[0]
    if len(hit) > 1:
        raise Exception('Repeated tau values')
    if len(hit) == 1:
        s = hit[0]
        seg = tau_r[s:s + len(omega)]
        subset = len(seg) == len(omega) and bool(np.all(seg == inv_om))
    if not subset:
        om_r = _rel_round(omega, 10)
        hit = np.where(inv_om == tau_r[0])[0]
        if len(hit) > 1:
            raise Exception('Repeated omega values')
        if len(hit) == 1:
            s = hit[0]
            seg = om_r[s:s + len(tau)]
            subset = len(seg) == len(tau) and bool(np.all(seg == _rel_round(1 / tau, 10)))
    if is_loguniform(frequencies) and not ct:
        return bool(tau_eq_omega or (subset and is_loguniform(tau)))
    return False


BASIS_IDS = {'gaussian': 0, 'Cole-Cole': 1, 'Zic': 2}


def construct_A(frequencies, part, tau=None, epsilon=1.0, kernel='DRT', dist_type='series', symmetry='planar',
                bc=None, ct=False, k_ct=None, toeplitz=None, basis='gaussian'):
    f = _f64(frequencies)
    tau = _f64(1 / (2 * np.pi * f) if tau is None else tau)
    if toeplitz is None:
        toeplitz = a_is_toeplitz(f, tau, ct)
    kid = KERNEL_IDS[('DRT', None, None)] if kernel == 'DRT' else KERNEL_IDS[('DDT', bc, symmetry)]
    out = np.empty((len(f), len(tau)))
    rc = lib().orc_build_A_basis(_p(f), len(f), _p(tau), len(tau), C.c_double(epsilon), kid,
                                 0 if part == 'real' else 1, int(dist_type == 'series'), int(bool(ct)),
                                 C.c_double(k_ct if k_ct is not None else 0.0), int(toeplitz), BASIS_IDS[basis], _p(out))
    if rc != 0:
        raise Exception('First entries of first row and column are not equal')
    return out


def _order_coefs(order, n):
    c = np.zeros(n)
    if isinstance(order, (list, tuple)):
        c[:3] = order
    elif order in (0, 1, 2, 3) and order < n:
        c[int(order)] = 1.0
    elif 0 < order < 1:
        c[0], c[1] = 1 - order, order
    elif 1 < order < 2:
        c[1], c[2] = 2 - order, order - 1
    else:
        raise ValueError('Order must be between 0 and 3')
    return c


def construct_L(tau, epsilon, order):
    tau = _f64(tau)
    out = np.empty((len(tau), len(tau)))
    lib().orc_build_L(_p(tau), len(tau), C.c_double(epsilon), _p(_order_coefs(order, 4)), _p(out))
    return out


def construct_L_rect(frequencies, tau, epsilon, order, basis='gaussian'):
    """construct_L for any frequencies against any tau (matrices.py:268-325): [len(frequencies) x len(tau)]."""
    f, tau = _f64(frequencies), _f64(tau)
    if basis == 'Zic' and order != 0:
        raise ValueError('the Zic basis has order 0 only (matrices.py:316-318)')
    out = np.empty((len(f), len(tau)))
    lib().orc_build_L_rect(_p(f), len(f), _p(tau), len(tau), C.c_double(epsilon), _p(_order_coefs(order, 4)),
                           BASIS_IDS[basis], _p(out))
    return out


def construct_M(tau, epsilon, order):
    tau = _f64(tau)
    out = np.empty((len(tau), len(tau)))
    toep = is_loguniform(1 / (2 * np.pi * tau))
    lib().orc_build_M(_p(tau), len(tau), C.c_double(epsilon), _p(_order_coefs(order, 3)), int(toep), _p(out))
    return out


# ------------------------------------------------------------------------------------------ NUTS oracle
class NutsControl(C.Structure):
    _fields_ = [('adapt_delta', C.c_double), ('adapt_t0', C.c_double), ('adapt_gamma', C.c_double),
                ('adapt_kappa', C.c_double), ('max_treedepth', C.c_int), ('init_buffer', C.c_int),
                ('term_buffer', C.c_int), ('base_window', C.c_int), ('init_radius', C.c_double),
                ('max_deltaH', C.c_double), ('stepsize0', C.c_double)]


class ChainDiag(C.Structure):
    _fields_ = [('n_leapfrog', C.c_longlong), ('n_divergent', C.c_int), ('n_max_treedepth', C.c_int),
                ('stepsize', C.c_double), ('mean_accept', C.c_double)]


def nuts_control(**kw):
    c = NutsControl()
    lib().orc_nuts_defaults(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _diag_dict(d):
    return dict(n_leapfrog=d.n_leapfrog, n_divergent=d.n_divergent, n_max_treedepth=d.n_max_treedepth,
                stepsize=d.stepsize, mean_accept=d.mean_accept)


def nuts_sample(model, chain_id, seed, warmup, n_draws, init_theta=None, control=None):
    """One chain of the recursive CPU NUTS on an OracleModel.  Returns (draws [n_draws x D], lp, diag)."""
    ctrl = control if control is not None else nuts_control()
    draws = np.empty((n_draws, model.D)); lp = np.empty(n_draws)
    diag = ChainDiag()
    init = _f64(init_theta) if init_theta is not None else None
    fn = lib().orc_nuts_sample
    fn.restype = C.c_int
    rc = fn(C.byref(model.m), int(chain_id), C.c_uint64(seed), int(warmup), int(n_draws),
            _p(init) if init is not None else None, C.byref(ctrl), _p(draws), _p(lp), C.byref(diag))
    if rc != 0:
        raise RuntimeError('orc_nuts_sample failed (no finite initial point)')
    return draws, lp, _diag_dict(diag)


def nuts_sample_gauss(mu, sd, chain_id, seed, warmup, n_draws, control=None):
    """Known-answer target N(mu, diag(sd^2)) through the same sampler."""
    mu = _f64(mu); sd = _f64(sd)
    desc = np.concatenate([[float(len(mu))], mu, sd])
    ctrl = control if control is not None else nuts_control()
    draws = np.empty((n_draws, len(mu)))
    diag = ChainDiag()
    fn = lib().orc_nuts_sample_gauss
    fn.restype = C.c_int
    rc = fn(_p(desc), int(chain_id), C.c_uint64(seed), int(warmup), int(n_draws), C.byref(ctrl), _p(draws),
            C.byref(diag))
    assert rc == 0
    return draws, _diag_dict(diag)


def eval_loop(model, theta0, n, jacobian=True):
    fn = lib().orc_eval_loop
    fn.restype = C.c_double
    return fn(C.byref(model.m), _p(_f64(theta0)), int(n), int(jacobian))


# ---- the tuned CPU evaluator of the headline family (oracle/bdrt_tuned.c): bench.py's cpu_baseline.tuned, checked against the oracle
_TUNED = None


def tuned_lib(force=False):
    global _TUNED
    so = os.path.join(_HERE, 'libtuned_native.so')
    src = os.path.join(_HERE, 'bdrt_tuned.c')
    if force and os.path.exists(so):
        os.remove(so); _TUNED = None
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libtuned_native.so'])
    if _TUNED is None:
        _TUNED = C.CDLL(so)
        _TUNED.tuned_s1_create.restype = C.c_void_p
        _TUNED.tuned_s1_logp_grad.restype = C.c_double
        _TUNED.tuned_s1_bench.restype = C.c_double
        _TUNED.tuned_s1_is_banded.restype = C.c_int
    return _TUNED


class TunedS1:
    """Series / Series_pos without outlier parameters on one DRT block: the oracle's log-posterior and gradient from a preallocated
    workspace, dense (banded=False) or through the diagonals of the penalty operators when they are banded Toeplitz."""

    def __init__(self, blk, Z, freq, sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0, banded=True):
        L = tuned_lib()
        self._a = [_f64(blk[k]) for k in ('A', 'L0', 'L1', 'L2')] + [_f64(Z), _f64(freq)]
        A = self._a[0]
        self.nf, self.K = A.shape[0] // 2, A.shape[1]
        self.D = 2 * self.K + 9
        self.h = C.c_void_p(L.tuned_s1_create(C.c_int(self.nf), C.c_int(self.K), C.c_int(int(bool(blk.get('nonneg', False)))),
                                              *[_p(a) for a in self._a[:5]], _p(self._a[5]), C.c_double(sigma_min), C.c_double(ups_alpha),
                                              C.c_double(ups_beta), C.c_double(induc_scale), C.c_int(int(banded))))
        self.banded = bool(L.tuned_s1_is_banded(self.h))

    def logp_grad(self, theta, jacobian=True):
        th = _f64(theta); g = np.empty(self.D)
        lp = tuned_lib().tuned_s1_logp_grad(self.h, _p(th), C.c_int(int(jacobian)), _p(g))
        return float(lp), g

    def bench(self, theta0, n):
        th = _f64(theta0); g = np.empty(self.D)
        return float(tuned_lib().tuned_s1_bench(self.h, _p(th), C.c_int(int(n)), _p(g)))

    def __del__(self):
        try:
            tuned_lib().tuned_s1_destroy(self.h)
        except Exception:
            pass
