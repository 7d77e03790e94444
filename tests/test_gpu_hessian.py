"""GPU tests (-m gpu) of the closed-form Hessian behind the MAP's Newton iteration (csrc/bdrt_newton_hess.h; StanModel.optimizing,
reference bayes_drt/inversion.py:1216; density stan_model_files/Series_pos_modelcode.txt:24-69): the HIP kernels against the numpy
statement of the same formulas (itself held to central differences of the oracle's gradient by tests/test_oracle_hessian.py) and
against central differences of the HIP gradient; the Newton iteration on it against the finite-difference iteration of rounds 1-5."""
import ctypes as C
import time

import numpy as np
import pytest

from tests.helpers import load, rel_l2
from tests.hessian_numpy import series_hessian

pytestmark = pytest.mark.gpu


def _dat_problem(name, pos=True, n_spectra=1):
    from bayes_drt_amd.model import Problem
    d = load(name)
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=pos)
    kw = dict(sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']), induc_scale=float(d['induc_scale']))
    Z = d['Z'] if n_spectra == 1 else np.vstack([d['Z'] * (1.0 + 0.1 * i) for i in range(n_spectra)])
    return Problem([blk], Z, d['freq'], **kw), d, kw


def _hip_hessian(prob, y, spec=0):
    H = np.empty((prob.D, prob.D))
    rc = prob._lib.bdrt_debug_hessian(C.c_void_p(prob.handle), y.ctypes.data_as(C.c_void_p), int(spec), H.ctypes.data_as(C.c_void_p))
    return rc, H


@pytest.mark.parametrize('name,pos', [('dat_optimize_2ZARC_uniform_0.25_K81', True), ('dat_optimize_2ZARC_uniform_0.25_K161', True),
                                      ('dat_optimize_2ZARC_uniform_0.25_K101', False), ('dat_sample_2ZARC_uniform_0.25_K161', True)])
def test_hip_hessian_equals_the_numpy_statement(name, pos):
    """Every entry of the D x D matrix, relative to its largest entry: <= 1e-11 (both sum the same terms in another order)."""
    prob, d, kw = _dat_problem(name, pos, n_spectra=3)
    rs = np.random.RandomState(5)
    for spec in (0, 2):
        y = np.ascontiguousarray(rs.uniform(-1.5, 1.5, prob.D))
        rc, H = _hip_hessian(prob, y, spec)
        assert rc == 0
        Z = d['Z'] * (1.0 + 0.1 * spec)
        kw2 = {k: v for k, v in kw.items() if k != 'induc_scale'}
        _, _, Href = series_hessian(y, d['A'], (d['L0'], d['L1'], d['L2']), Z, 2 * np.pi * d['freq'], pos=pos, induc_scale=kw['induc_scale'], **kw2)
        sc = np.max(np.abs(Href))
        low = np.tril(np.ones_like(H, dtype=bool))          # (the factorisation reads the lower triangle)
        assert np.max(np.abs(H - Href)[low]) <= 1e-11 * sc, np.max(np.abs(H - Href)) / sc
        assert np.max(np.abs(H - H.T)) <= 1e-12 * sc
    prob.close()


def test_hip_hessian_on_a_general_frequency_grid_equals_central_differences_of_the_hip_gradient():
    """A measured spectrum on its own irregular frequency list: A is a general matrix (plain copies instead of Toeplitz generators)."""
    from bayes_drt_amd import matrices as gm
    from bayes_drt_amd.model import Problem
    rs = np.random.RandomState(3)
    nf, K = 37, 61
    f = np.sort(10 ** rs.uniform(-1.5, 5, nf))[::-1].copy()
    bf = np.logspace(6, -2.5, K)
    tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    w = 2 * np.pi * f
    z = 1.0 + 1.2 / (1 + (1j * w * 3e-3) ** 0.8) + 0.003 * (rs.normal(size=nf) + 1j * rs.normal(size=nf))
    z = z / (np.std(np.abs(z)) / np.sqrt(nf / 81))
    blk = dict(A=A, L0=0.36 * L[0], L1=0.24 * L[1], L2=0.12 * L[2], nonneg=True)
    prob = Problem([blk], np.concatenate([z.real, z.imag]), f, sigma_min=0.002, ups_alpha=0.05, ups_beta=0.1)
    y = np.ascontiguousarray(rs.uniform(-1, 1, prob.D))
    rc, H = _hip_hessian(prob, y)
    assert rc == 0
    Hfd = np.zeros_like(H)
    for j in range(prob.D):
        h = 1e-6 * max(1.0, abs(y[j]))
        P_ = np.vstack([y, y]); P_[0, j] += h; P_[1, j] -= h
        _, g = prob.logp_grad(P_, jacobian=False)
        Hfd[j] = (g[0] - g[1]) / (2 * h)
    Hfd = 0.5 * (Hfd + Hfd.T)
    assert np.max(np.abs(H - Hfd)) <= 1e-7 * np.max(np.abs(Hfd))
    prob.close()


def test_models_without_a_closed_form_say_so():
    from tests.test_gpu_engine import _small_problem
    from bayes_drt_amd.model import Problem
    blk, Z, f, kw = _small_problem()                      # 12 x 9: L is not banded at this size
    prob = Problem([blk], Z, f, **kw)
    rc, _ = _hip_hessian(prob, np.zeros(prob.D))
    assert rc == 1
    prob.close()


@pytest.mark.parametrize('tag', ['K81', 'K161'])
def test_newton_iteration_on_the_closed_form_hessian(tag, monkeypatch):
    """The same stationary point as the finite-difference iteration (gamma to 1e-6), with a handful of evaluations per round
    instead of 2 D + 4: VERDICT r05's bound of 2000 gradient evaluations per K = 161 fit."""
    from bayes_drt_amd.engine import optimize_batch
    prob, d, kw = _dat_problem('dat_optimize_2ZARC_uniform_0.25_%s' % tag)
    th0 = np.random.RandomState(1234).uniform(-2, 2, (1, prob.D))
    t0 = time.time(); out, rep = optimize_batch(prob, th0); t_an = time.time() - t0
    monkeypatch.setenv('BDRT_NEWTON_FD', '1')
    t0 = time.time(); out_fd, rep_fd = optimize_batch(prob, th0); t_fd = time.time() - t0
    K = prob.Ks[0]
    print('%s: closed form %d Newton rounds, %d evaluations, %.1f ms; finite differences %d rounds, %d evaluations, %.1f ms; lp %.6f / %.6f'
          % (tag, rep[0]['newton_iterations'], rep[0]['n_evals'], 1e3 * t_an, rep_fd[0]['newton_iterations'], rep_fd[0]['n_evals'], 1e3 * t_fd,
             rep[0]['lp'], rep_fd[0]['lp']))
    assert rep[0]['return_code'] == 0 and rep[0]['grad_inf'] < 1e-8
    assert rep[0]['n_evals'] <= 2000 and rep_fd[0]['n_evals'] > 20000
    assert abs(rep[0]['lp'] - rep_fd[0]['lp']) <= 1e-8 * abs(rep_fd[0]['lp'])
    assert rel_l2(np.exp(out[0][2:2 + K]), np.exp(out_fd[0][2:2 + K])) < 1e-6
    prob.close()
