"""GPU inference engine with the interface the reference expects from pystan's StanModel.

`Inverter.fit` (reference bayes_drt/inversion.py:1190-1221) obtains an object from `_get_stan_model` and calls
    model.optimizing(dat, iter=max_iter, seed=random_seed, init=init)                         -> mapping name -> ndarray
    model.sampling(dat, warmup=, iter=, chains=, seed=, init=, control={'adapt_delta','adapt_t0'})  -> fit[name] -> draws
`StanModel` below provides exactly that, with every log-posterior / gradient evaluated by libbdrt.so on the GPU:
L-BFGS (bdrt_optimize) for `optimizing`, the device-resident NUTS (bdrt_sampler_*) for `sampling`.
There is no CPU path.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import ChainDiag, NutsControl, OptOptions, OptReport, check, f64, ptr
from .model import Problem

# model registry: same file names as the reference's stan_models.model_dict (bayes_drt/stan_models.py:20-38)
MODEL_NAMES = [
    'Series_StanModel.pkl', 'Series_outliers_StanModel.pkl', 'Series_pos_StanModel.pkl',
    'Series_pos_outliers_StanModel.pkl', 'Parallel_StanModel.pkl',
    'Series-Parallel_StanModel.pkl', 'Series-Parallel_pos_StanModel.pkl',
    'Series-Parallel_pos_outliers_StanModel.pkl', 'Series-Parallel_outliers_StanModel.pkl',
    'Series-2Parallel_StanModel.pkl', 'Series-2Parallel_pos_StanModel.pkl',
]
# shipped by the reference but out of scope here:
UNSUPPORTED = {'Parallel_outliers_StanModel.pkl': 'size-inconsistent Stan program in the reference (SURVEY fact 9)',
               'Parallel_fitY_StanModel.pkl': 'experimental admittance fit ("for testing only", inversion.py:1150)',
               'Parallel_fitY_SA_StanModel.pkl': 'experimental admittance fit ("for testing only", inversion.py:1150)'}


def _family(model_name):
    base = model_name.replace('_StanModel.pkl', '')
    fam = base.split('_')[0]
    return fam, '_pos' in base, base.endswith('_outliers')


def blocks_from_dat(model_name, dat):
    """Translate the reference's Stan data dict (inversion.py:1739-1754, :1929-1955, :2012-2042) into blocks."""
    fam, pos, outl = _family(model_name)
    if fam == 'Series':
        blocks = [dict(A=dat['A'], L0=dat['L0'], L1=dat['L1'], L2=dat['L2'], parallel=False, nonneg=pos)]
        names = dict(x=['x'], raw=['x'], ups=['ups_raw'], d=[('d0_strength', 'd1_strength', 'd2_strength')],
                     q=['q'], upsT=['ups'], dups=['dups'])
    elif fam == 'Parallel':
        blocks = [dict(A=dat['A'], L0=dat['L0'], L1=dat['L1'], L2=dat['L2'], parallel=True, nonneg=True, x_scale=1.0)]
        names = dict(x=['x'], raw=['x'], ups=['ups_raw'], d=[('d0_strength', 'd1_strength', 'd2_strength')],
                     q=['q'], upsT=['ups'], dups=['dups'])
    elif fam == 'Series-Parallel':
        blocks = [dict(A=dat['As'], L0=dat['L0s'], L1=dat['L1s'], L2=dat['L2s'], parallel=False, nonneg=pos),
                  dict(A=dat['Ap'], L0=dat['L0p'], L1=dat['L1p'], L2=dat['L2p'], parallel=True, nonneg=True,
                       x_scale=float(dat['xp_scale']))]
        names = dict(x=['xs', 'xp'], raw=['xs', 'xp_raw'], ups=['ups_s_raw', 'ups_p_raw'],
                     d=[('d0s_strength', 'd1s_strength', 'd2s_strength'), ('d0p_strength', 'd1p_strength', 'd2p_strength')],
                     q=['qs', 'qp'], upsT=['ups_s', 'ups_p'], dups=['dups_s', 'dups_p'])
    elif fam == 'Series-2Parallel':
        blocks = [dict(A=dat['As'], L0=dat['L0s'], L1=dat['L1s'], L2=dat['L2s'], parallel=False, nonneg=pos),
                  dict(A=dat['Ap1'], L0=dat['L0p1'], L1=dat['L1p1'], L2=dat['L2p1'], parallel=True, nonneg=True,
                       x_scale=float(dat['xp1_scale'])),
                  dict(A=dat['Ap2'], L0=dat['L0p2'], L1=dat['L1p2'], L2=dat['L2p2'], parallel=True, nonneg=True,
                       x_scale=float(dat['xp2_scale']))]
        names = dict(x=['xs', 'xp1', 'xp2'], raw=['xs', 'xp1_raw', 'xp2_raw'],
                     ups=['ups_s_raw', 'ups_p1_raw', 'ups_p2_raw'],
                     d=[tuple('d%d%s_strength' % (i, s) for i in range(3)) for s in ('s', 'p1', 'p2')],
                     q=['qs', 'qp1', 'qp2'], upsT=['ups_s', 'ups_p1', 'ups_p2'], dups=['dups_s', 'dups_p1', 'dups_p2'])
    else:
        raise ValueError('No GPU model for %s' % model_name)
    kw = dict(sigma_min=float(dat['sigma_min']), ups_alpha=float(dat['ups_alpha']), ups_beta=float(dat['ups_beta']),
              induc_scale=float(dat['induc_scale']), use_x_sum=(len(blocks) > 1),
              x_sum_invscale=float(dat.get('x_sum_invscale', 0.0)))
    if outl:
        if fam == 'Series':
            # package form: sigma_out_raw[N], sigma_out_scale[N] (Series_outliers_modelcode.txt:30-31,45,71-72)
            kw.update(outlier_mode=1, so_lambda=float(dat['sigma_out_lambda']), so_alpha=float(dat['sigma_out_alpha']),
                      so_beta=float(dat['sigma_out_beta']))
        else:
            # stacked form, N = 2*Nf (Series-Parallel_pos_outliers_modelcode.txt:42,64,107)
            kw.update(outlier_mode=2, so_lambda=float(dat['so_invscale']))
    return blocks, kw, names


def _report(r):
    return dict(iterations=r.iterations, n_evals=r.n_evals, return_code=r.return_code, lp=r.lp, grad_norm=r.grad_norm,
                newton_iterations=r.newton_iterations, grad_inf=r.grad_inf)


class StanFit:
    """What `fit[name]` needs (reference inversion.py:2514-2519, :2560, :2702, :3096): post-warm-up draws of all
    chains merged, shape [chains*draws, ...]."""

    def __init__(self, model, theta, lp, diag, chains, n_draws):
        self._model = model
        self.theta = theta                      # [chains*draws, D] unconstrained
        self.lp = lp
        self.diagnostics = diag
        self.chains, self.n_draws = chains, n_draws
        self._params = model.problem.constrain(theta)
        self._cache = {}
        self.stepsize = [d['stepsize'] for d in diag]
        self.n_leapfrog = int(sum(d['n_leapfrog'] for d in diag))
        self.n_divergent = int(sum(d['n_divergent'] for d in diag))
        self.n_max_treedepth = int(sum(d['n_max_treedepth'] for d in diag))

    def _transformed(self):
        if 'Z_hat' not in self._cache:
            _, Zh, sg = self._model.problem.transformed(self.theta)
            self._cache['Z_hat'], self._cache['sigma_tot'] = Zh, sg
        return self._cache

    def __getitem__(self, name):
        if name == 'lp__':
            return self.lp
        if name in ('Z_hat', 'sigma_tot'):
            return self._transformed()[name]
        return self._model._extract(self._params, name)

    def keys(self):
        return self._model.param_names()

    def chain_draws(self, name):
        """[chains, draws, ...] view for convergence diagnostics."""
        a = self[name]
        return a.reshape((self.chains, self.n_draws) + a.shape[1:])

    def to_saved(self):
        """Plain-array snapshot for `Inverter.save_fit_data`: no GPU handle, loadable anywhere."""
        data = {k: np.array(self[k]) for k in self.keys()}
        data['lp__'] = np.array(self.lp)
        return SavedFit(data, self.chains, self.n_draws, self.diagnostics, np.array(self.theta))


class SavedFit(dict):
    """HMC result restored from a file: the same `fit[name]` / `chain_draws` / diagnostics surface as StanFit, backed by
    stored arrays only (what the reference keeps as a pickled pystan fit object, inversion.py:3990)."""

    def __init__(self, data=(), chains=1, n_draws=0, diagnostics=(), theta=None):
        super().__init__(data)
        self.chains, self.n_draws = int(chains), int(n_draws)
        self.diagnostics = list(diagnostics)
        self.theta = theta
        self.lp = self.get('lp__')
        self.stepsize = [d['stepsize'] for d in self.diagnostics]
        self.n_leapfrog = int(sum(d['n_leapfrog'] for d in self.diagnostics))
        self.n_divergent = int(sum(d['n_divergent'] for d in self.diagnostics))
        self.n_max_treedepth = int(sum(d['n_max_treedepth'] for d in self.diagnostics))

    def chain_draws(self, name):
        a = self[name]
        return a.reshape((self.chains, self.n_draws) + a.shape[1:])

    def to_saved(self):
        return self

    def __reduce__(self):
        return (SavedFit, (dict(self), self.chains, self.n_draws, self.diagnostics, self.theta))


class StanModel:
    """GPU-backed stand-in for the compiled pystan model `_get_stan_model` returns."""

    def __init__(self, model_name):
        if model_name in UNSUPPORTED:
            raise NotImplementedError('%s: %s' % (model_name, UNSUPPORTED[model_name]))
        if model_name not in MODEL_NAMES:
            raise ValueError('Unknown model %s' % model_name)
        self.model_name = model_name
        self.problem = None
        self._names = None
        self.last_report = None

    # ------------------------------------------------------------------ problem set-up
    def _prepare(self, dat):
        blocks, kw, names = blocks_from_dat(self.model_name, dat)
        Z = f64(dat['Z'])
        fam, pos, outl = _family(self.model_name)
        nf = len(dat['freq'])
        if outl and fam == 'Series' and int(dat['N']) != nf:
            raise ValueError('Series outlier models take N = number of frequencies (inversion.py:1208-1211)')
        self.problem = Problem(blocks, Z, dat['freq'], **kw)
        self._names = names
        self._lay = self.problem.layout()
        return self.problem

    def param_names(self):
        n = self._names
        out = ['Rinf_raw', 'induc_raw'] + n['raw'] + ['sigma_res_raw', 'alpha_prop_raw', 'alpha_re_raw', 'alpha_im_raw']
        if self.problem.dat.outlier_mode:
            out += ['sigma_out_raw'] + (['sigma_out_scale'] if self.problem.dat.outlier_mode == 1 else [])
        out += n['ups'] + [k for t in n['d'] for k in t]
        out += ['Rinf', 'induc', 'sigma_res', 'alpha_prop', 'alpha_re', 'alpha_im', 'Z_hat', 'sigma_tot'] + n['upsT']
        out += [x for x, r in zip(n['x'], n['raw']) if x != r]
        if self.problem.dat.outlier_mode:
            out.append('sigma_out')
        return out

    def _extract(self, params, name):
        """name -> array from constrained parameter rows [B x D] (transformed parameters by their Stan definitions)."""
        P, lay, n = self.problem, self._lay, self._names
        nf = P.nf
        one = params.ndim == 1
        p = np.atleast_2d(params)

        def ret(a):
            return a[0] if one else a
        if name == 'Rinf_raw': return ret(p[:, 0])
        if name == 'induc_raw': return ret(p[:, 1])
        if name == 'Rinf': return ret(100.0 * p[:, 0])
        if name == 'induc': return ret(p[:, 1] * P.dat.induc_scale)
        err = {'sigma_res': 0, 'alpha_prop': 1, 'alpha_re': 2, 'alpha_im': 3}
        if name in err: return ret(0.05 * p[:, lay['err'] + err[name]])
        if name.endswith('_raw') and name[:-4] in err: return ret(p[:, lay['err'] + err[name[:-4]]])
        for b, K in enumerate(P.Ks):
            xs = p[:, lay['x'][b]:lay['x'][b] + K]
            if name == n['raw'][b]: return ret(xs)
            if name == n['x'][b]: return ret(xs * P.dat.x_scale[b] if P.dat.is_parallel[b] else xs)
            us = p[:, lay['ups'][b]:lay['ups'][b] + K]
            if name == n['ups'][b]: return ret(us)
            if name == n['upsT'][b]: return ret(0.15 * us)
            if name == n['dups'][b]:
                u = 0.15 * us
                return ret(0.5 * (u[:, 1:-1] - 0.5 * (u[:, :-2] + u[:, 2:])) / u[:, 1:-1])
            for i, dn in enumerate(n['d'][b]):
                if name == dn: return ret(p[:, lay['d'][b] + i])
        if P.dat.outlier_mode:
            so = lay['so']
            if name == 'sigma_out_raw':
                return ret(p[:, so:so + (nf if P.dat.outlier_mode == 1 else 2 * nf)])
            if name == 'sigma_out_scale' and P.dat.outlier_mode == 1: return ret(p[:, so + nf:so + 2 * nf])
            if name == 'sigma_out':
                if P.dat.outlier_mode == 1: return ret(0.05 * p[:, so:so + nf] * p[:, so + nf:so + 2 * nf])
                return ret(0.05 * p[:, so:so + 2 * nf])
        raise KeyError(name)

    # ------------------------------------------------------------------ initial values
    def _init_theta(self, init, n, seed):
        """Stan semantics: 'random' -> U(-2,2) on the unconstrained scale; a dict / callable gives constrained values
        for some parameters (inversion.py:1654-1682: x, Rinf_raw, induc_raw[, sigma_out_raw]; other keys ignored), the
        rest random.  The stream is numpy's (Stan's own RNG is not reproducible)."""
        P = self.problem
        rs = np.random.RandomState(seed)
        theta = rs.uniform(-2, 2, (n, P.D))
        if init is None or (isinstance(init, str) and init == 'random'):
            return theta
        if isinstance(init, (int, float)) and init == 0:
            return np.zeros((n, P.D))
        inits = [init() if callable(init) else init for _ in range(n)] if not isinstance(init, (list, tuple)) else list(init)
        lay, names = self._lay, self._names
        slots = {'Rinf_raw': (0, 1), 'induc_raw': (1, 1)}
        for b, K in enumerate(P.Ks):
            slots[names['raw'][b]] = (lay['x'][b], K)
            slots[names['ups'][b]] = (lay['ups'][b], K)
        if P.dat.outlier_mode == 1:
            slots['sigma_out_raw'] = (lay['so'], P.nf)
            slots['sigma_out_scale'] = (lay['so'] + P.nf, P.nf)
        elif P.dat.outlier_mode == 2:
            slots['sigma_out_raw'] = (lay['so'], 2 * P.nf)
        for i, iv in enumerate(inits):
            for key, val in iv.items():
                if key not in slots:
                    continue
                o, k = slots[key]
                v = np.broadcast_to(np.asarray(val, dtype=float).ravel(), (k,)) if np.ndim(val) else np.full(k, float(val))
                pos = P.is_pos[o:o + k]
                if np.any(pos & (v <= 0)):
                    raise ValueError('init value for %s must be positive (declared lower=0)' % key)
                theta[i, o:o + k] = np.where(pos, np.log(np.where(pos, v, 1.0)), v)
        return theta

    # ------------------------------------------------------------------ optimizing / sampling
    def optimizing(self, data, iter=50000, seed=1234, init='random', algorithm='LBFGS+Newton', extra_inits=(), **opts):
        """MAP without Jacobian (Stan `optimizing`).  Returns OrderedDict name -> ndarray with the parameters and
        transformed parameters `Inverter._extract_parameter` reads.

        extra_inits: further starting points (Stan `init=` values).  All starts run as one lock-step batch of
        `bdrt_optimize` (the wall time of the slowest one); the answer is the start `init` unless another start ends at a
        clearly higher log-posterior (`last_report['start']` tells which, `last_report['starts']` lists all of them).

        algorithm='LBFGS' is the Stan-style L-BFGS(5) with Stan's termination tests (an early-terminated iterate,
        SURVEY fact 4); the default 'LBFGS+Newton' continues with the GPU full-Hessian Newton polish to a true
        stationary point (bdrt_newton.h)."""
        P = self._prepare(data)
        lib = P._lib
        rows = [self._init_theta(init, 1, seed)]
        for e in extra_inits:
            try:
                if isinstance(e, tuple) and e[0] == 'random':           # ('random', k): another draw of the random start
                    rows.append(self._init_theta('random', 1, seed + int(e[1])))
                    continue
                rows.append(self._init_theta(e, 1, seed))
            except ValueError:                          # a candidate outside the support is simply not used
                pass
        theta0 = np.vstack(rows)
        n = theta0.shape[0]
        o = OptOptions()
        lib.bdrt_opt_defaults(C.byref(o))
        o.max_iter = int(iter)
        if algorithm == 'LBFGS':
            o.newton_max_iter = 0
        elif algorithm != 'LBFGS+Newton':
            raise ValueError("algorithm must be 'LBFGS' or 'LBFGS+Newton'")
        for k, v in opts.items():
            setattr(o, k, v)
        out = np.empty((n, P.D))
        rep = (OptReport * n)()
        check(lib.bdrt_optimize(P.handle, ptr(theta0), None, n, C.byref(o), ptr(out), rep), 'bdrt_optimize')
        reports = [_report(r) for r in rep]
        best = 0
        for i in range(1, n):
            a, b = reports[i], reports[best]
            higher = np.isfinite(a['lp']) and (not np.isfinite(b['lp']) or a['lp'] > b['lp'] + 1e-6 * max(1.0, abs(b['lp'])))
            if higher and (a['return_code'] == 0 or b['return_code'] != 0):
                best = i
        self.last_report = dict(reports[best], start=best, starts=reports)
        return self.result_dict(out[best])

    def result_dict(self, theta):
        P = self.problem
        params, Zh, sg = P.transformed(np.atleast_2d(theta))
        res = OrderedDict()
        for name in self.param_names():
            if name == 'Z_hat':
                res[name] = Zh[0]
            elif name == 'sigma_tot':
                res[name] = sg[0]
            else:
                res[name] = self._extract(params[0], name)
        nf = P.nf
        res['Z_hat_re'] = np.concatenate([Zh[0][:nf], Zh[0][:nf]])
        res['Z_hat_im'] = np.concatenate([Zh[0][nf:], Zh[0][nf:]])
        res['theta_unconstrained'] = np.asarray(theta).copy()
        return res

    def sampling(self, data, warmup=200, iter=400, chains=2, seed=1234, init='random', control=None,
                 chain_ids=None, rounds_per_launch=None):
        """NUTS with Jacobian (Stan `sampling`): `iter` counts warm-up + draws, like pystan."""
        P = self._prepare(data)
        n_draws = int(iter) - int(warmup)
        if n_draws < 0:
            raise ValueError('iter must be >= warmup')
        ctrl = NutsControl()
        P._lib.bdrt_nuts_defaults(C.byref(ctrl))
        for k, v in (control or {}).items():
            setattr(ctrl, k, v)
        init_theta = None
        if not (init is None or (isinstance(init, str) and init == 'random')):
            init_theta = self._init_theta(init, chains, seed)
        draws, lp, diag = sample_units(P, chains, warmup, n_draws, seed, ctrl, init_theta=init_theta,
                                       chain_ids=chain_ids, rounds_per_launch=rounds_per_launch)
        theta = draws.reshape(chains * n_draws, P.D)
        return StanFit(self, theta, lp.reshape(-1), diag, chains, n_draws)


class DeviceArray:
    """Zero-copy view of a device buffer for consumers that understand `__cuda_array_interface__` (torch on ROCm does):
    lets torch.distributed (RCCL) gather the sampler's draws straight from HBM."""

    def __init__(self, ptr_value, shape, owner):
        self._owner = owner                      # keeps the sampler (and its device memory) alive
        self.__cuda_array_interface__ = {'shape': tuple(int(x) for x in shape), 'typestr': '<f8',
                                         'data': (int(ptr_value), False), 'version': 2, 'strides': None}


class Sampler:
    """A set of device-resident NUTS chains (bdrt_sampler of include/bdrt.h): unit u samples spectrum spec[u] with the
    RNG stream (seed, chain_ids[u]).  The draws stay in HBM until `results()`; `summary()` reduces them there."""

    def __init__(self, problem, n_units, warmup, n_draws, seed, ctrl=None, spec=None, chain_ids=None, init_theta=None):
        lib = problem._lib
        if ctrl is None:
            ctrl = NutsControl()
            lib.bdrt_nuts_defaults(C.byref(ctrl))
        sp = None if spec is None else np.ascontiguousarray(np.asarray(spec, dtype=np.int32))
        ci = None if chain_ids is None else np.ascontiguousarray(np.asarray(chain_ids, dtype=np.int32))
        it = None if init_theta is None else f64(init_theta)
        self.problem, self._lib, self.ctrl = problem, lib, ctrl
        self.n_units, self.warmup, self.n_draws = int(n_units), int(warmup), int(n_draws)
        self.handle = lib.bdrt_sampler_create(problem.handle, self.n_units, ptr(sp), ptr(ci), self.warmup, self.n_draws,
                                              C.c_uint64(int(seed)), ptr(it), C.byref(ctrl))
        if not self.handle:
            raise _lib.BdrtError('bdrt_sampler_create: ' + lib.bdrt_last_error().decode())

    def close(self):
        if getattr(self, 'handle', None):
            self._lib.bdrt_sampler_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def advance(self, rounds, want_done=False):
        done = C.c_int(0)
        check(self._lib.bdrt_sampler_advance(self.handle, int(rounds), C.byref(done) if want_done else None),
              'bdrt_sampler_advance')
        return bool(done.value)

    def sync(self):
        check(self._lib.bdrt_sampler_sync(self.handle), 'bdrt_sampler_sync')

    def run(self, rounds_per_launch=None):
        if not rounds_per_launch:
            check(self._lib.bdrt_sampler_run(self.handle), 'bdrt_sampler_run')
            return
        bound = ((1 << self.ctrl.max_treedepth) + 64) * (self.warmup + self.n_draws + 2) + 200
        spent = 0
        while spent <= bound:
            if self.advance(rounds_per_launch, want_done=True):
                return
            spent += int(rounds_per_launch)
        raise _lib.BdrtError('sampler did not finish within the leapfrog bound')

    def total_leapfrogs(self):
        return int(self._lib.bdrt_sampler_total_leapfrogs(self.handle))

    def kind(self):
        """0: sixteen chains per workgroup, 1: one chain per workgroup with its state in LDS, 2: one chain per workgroup,
        general block model, 3: one chain per wavefront (2.5 ... 8 live chains per CU of the single-DRT family), 4: one chain per
        workgroup on the streamed evaluator (problem beyond the LDS budget)."""
        return int(self._lib.bdrt_sampler_kind(self.handle))

    def tail_units(self):
        """Chains that `run` handed from the 16-chain kernel to the one-chain-per-workgroup kernel for the tail (0: none)."""
        return int(self._lib.bdrt_sampler_tail_units(self.handle))

    def compactions(self):
        """How often `run` re-packed the live chains into fewer 16-chain workgroups (runs with more than 16 units per CU)."""
        return int(self._lib.bdrt_sampler_compactions(self.handle))

    def kernel_time(self, reset=False):
        ms = C.c_double(); nl = C.c_int64()
        check(self._lib.bdrt_sampler_kernel_time(self.handle, C.byref(ms), C.byref(nl), int(reset)), 'bdrt_sampler_kernel_time')
        return ms.value, int(nl.value)

    def results(self, want_draws=True):
        """(draws [n_units, n_draws, D] unconstrained or None, lp [n_units, n_draws], per-chain diagnostics)."""
        draws = np.empty((self.n_units, self.n_draws, self.problem.D)) if want_draws else None
        lp = np.empty((self.n_units, self.n_draws))
        diag = (ChainDiag * self.n_units)()
        check(self._lib.bdrt_sampler_results(self.handle, ptr(draws), ptr(lp), diag), 'bdrt_sampler_results')
        dl = [dict(n_leapfrog=d.n_leapfrog, n_divergent=d.n_divergent, n_max_treedepth=d.n_max_treedepth,
                   stepsize=d.stepsize, mean_accept=d.mean_accept) for d in diag]
        if any(d['n_leapfrog'] < 0 for d in dl):
            raise _lib.BdrtError('a chain found no finite initial point in 100 attempts')
        return draws, lp, dl

    def draws_device(self):
        """The draws where the sampler left them (HBM), as a `__cuda_array_interface__` object."""
        p = self._lib.bdrt_sampler_draws_dev(self.handle)
        if not p:
            raise _lib.BdrtError('bdrt_sampler_draws_dev failed')
        return DeviceArray(p, (self.n_units, self.n_draws, self.problem.D), self)

    def summary(self, unit_lo, unit_hi, q=(2.5, 50.0, 97.5)):
        """Posterior mean [D] and percentiles [len(q), D] of the CONSTRAINED parameters over all draws of units
        [unit_lo, unit_hi), reduced on the device (np.mean / np.percentile of the reference, inversion.py:2517-2519, :2560)."""
        qa = np.ascontiguousarray(np.atleast_1d(np.asarray(q, dtype=np.float64)))
        mean = np.empty(self.problem.D); pct = np.empty((qa.size, self.problem.D))
        check(self._lib.bdrt_sampler_summary(self.handle, int(unit_lo), int(unit_hi), ptr(qa), qa.size, ptr(mean), ptr(pct)),
              'bdrt_sampler_summary')
        return mean, pct


def sample_units(problem, n_units, warmup, n_draws, seed, ctrl=None, spec=None, chain_ids=None, init_theta=None,
                 rounds_per_launch=None):
    """Run n_units chains (unit u: spectrum spec[u], RNG stream (seed, chain_ids[u])) to completion on the GPU.
    Returns draws [n_units, n_draws, D] (unconstrained), lp [n_units, n_draws], list of per-chain diagnostics."""
    with Sampler(problem, n_units, warmup, n_draws, seed, ctrl, spec=spec, chain_ids=chain_ids, init_theta=init_theta) as smp:
        smp.run(rounds_per_launch)
        return smp.results()


def optimize_batch(problem, theta0, spec=None, max_iter=50000, **opts):
    """Lock-step L-BFGS for several fits (rows of theta0; spectrum spec[i]).  Returns (theta [n x D], reports)."""
    lib = problem._lib
    theta0 = np.atleast_2d(f64(theta0))
    n = theta0.shape[0]
    o = OptOptions()
    lib.bdrt_opt_defaults(C.byref(o))
    o.max_iter = int(max_iter)
    for k, v in opts.items():
        setattr(o, k, v)
    sp = None if spec is None else np.ascontiguousarray(np.asarray(spec, dtype=np.int32))
    out = np.empty((n, problem.D))
    rep = (OptReport * n)()
    check(lib.bdrt_optimize(problem.handle, ptr(theta0), ptr(sp), n, C.byref(o), ptr(out), rep), 'bdrt_optimize')
    reports = [_report(r) for r in rep]
    return out, reports
