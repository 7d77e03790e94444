"""GPU tests (-m gpu) of the two inference loops: device-resident NUTS vs the recursive CPU oracle NUTS, and the
lock-step L-BFGS fed by GPU evaluations vs the same state machine fed by oracle evaluations."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import load, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_problem():
    """12 frequencies x 9 basis functions (general, non-Toeplitz matrices from the reference): D = 27."""
    m = load('mat_drt_general_12x9')
    A = np.vstack([m['A_re'], m['A_im']])
    blk = dict(A=A, L0=m['L0'], L1=m['L1'], L2=0.75 * m['L2'], nonneg=True)
    rng = np.random.default_rng(0)
    x_true = np.exp(-0.5 * ((np.log(m['tau']) + 4) / 2.0) ** 2)
    Z = A @ x_true + np.concatenate([np.full(12, 0.7), 1e-6 * 2 * np.pi * m['freq']])
    Z = Z / np.std(np.hypot(Z[:12], Z[12:])) + 0.01 * rng.standard_normal(24)
    kw = dict(sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0)
    return blk, Z, m['freq'], kw


def _bench_problem(mode='sample', tag='K161'):
    d = load('dat_%s_2ZARC_uniform_0.25_%s' % (mode, tag))
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    kw = dict(sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']),
              induc_scale=float(d['induc_scale']))
    return blk, d['Z'], d['freq'], kw, d


def _ctrl(lib, **kw):
    from bayes_drt_amd._lib import NutsControl
    c = NutsControl(); lib.bdrt_nuts_defaults(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_nuts_matches_oracle_draw_by_draw_small():
    """Same Philox streams, different program structure (checkpointed iteration on the GPU, recursion on the CPU):
    identical decisions, draws equal up to fp64 summation-order noise amplified by the dynamics."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    from oracle import oracle as orc
    blk, Z, f, kw = _small_problem()
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    warm, nd = 40, 25
    ctrl = _ctrl(prob._lib, max_treedepth=6)
    draws, lp, diag = sample_units(prob, 3, warm, nd, 2024, ctrl)
    n_match = 0
    for c in range(3):
        ref, lpr, dr = orc.nuts_sample(om, c, 2024, warm, nd, control=orc.nuts_control(max_treedepth=6))
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])      # identical tree shapes
        assert dr['n_divergent'] == diag[c]['n_divergent']
        assert abs(dr['stepsize'] - diag[c]['stepsize']) < 1e-6 * dr['stepsize']
        err = np.max(np.abs(draws[c] - ref), axis=1) / np.max(np.abs(ref))
        assert err[0] < 1e-6 and np.median(err) < 1e-5, err
        assert np.allclose(lp[c], lpr, rtol=1e-5, atol=1e-5)
        n_match += int(np.sum(err < 1e-4))
    assert n_match >= 0.9 * 3 * nd


def test_nuts_matches_oracle_benchmark_shape_short():
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    from oracle import oracle as orc
    blk, Z, f, kw, d = _bench_problem()
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    ctrl = _ctrl(prob._lib, max_treedepth=5)
    draws, lp, diag = sample_units(prob, 2, 6, 4, 1234, ctrl)
    for c in range(2):
        ref, lpr, dr = orc.nuts_sample(om, c, 1234, 6, 4, control=orc.nuts_control(max_treedepth=5))
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog']
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref))
        assert np.allclose(lp[c], lpr, rtol=1e-8, atol=1e-6)


def test_nuts_is_independent_of_packing_and_launch_slicing():
    """Draws are a function of (seed, chain id) only: not of which workgroup / column a chain lands in, nor of
    how the run is cut into kernel launches."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    blk, Z, f, kw = _small_problem()
    prob = Problem([blk], Z, f, **kw)
    ctrl = _ctrl(prob._lib, max_treedepth=5)
    ids = np.arange(20, dtype=np.int32)
    full, _, _ = sample_units(prob, 20, 20, 10, 7, ctrl, chain_ids=ids)
    part, _, _ = sample_units(prob, 3, 20, 10, 7, ctrl, chain_ids=ids[[17, 2, 9]])
    assert np.array_equal(part, full[[17, 2, 9]])
    sliced, _, _ = sample_units(prob, 3, 20, 10, 7, ctrl, chain_ids=ids[[17, 2, 9]], rounds_per_launch=7)
    assert np.array_equal(sliced, part)
    other, _, _ = sample_units(prob, 3, 20, 10, 8, ctrl, chain_ids=ids[[17, 2, 9]])
    assert not np.array_equal(other, part)


def test_nuts_posterior_matches_oracle_statistically():
    """Longer runs: GPU chains and oracle chains (different chain ids => independent streams) agree on the posterior
    mean of every parameter within Monte-Carlo error."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    from oracle import oracle as orc
    blk, Z, f, kw = _small_problem()
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    warm, nd = 300, 400
    g, _, dg = sample_units(prob, 16, warm, nd, 5, chain_ids=np.arange(100, 116, dtype=np.int32))
    o = np.stack([orc.nuts_sample(om, c, 5, warm, nd)[0] for c in range(4)])
    gm, om_ = g.reshape(-1, prob.D), o.reshape(-1, prob.D)
    sd = om_.std(axis=0)
    # conservative effective sample sizes: 1/10 of the draws
    se = sd * np.sqrt(1.0 / (gm.shape[0] / 10) + 1.0 / (om_.shape[0] / 10))
    z = np.abs(gm.mean(axis=0) - om_.mean(axis=0)) / se
    assert np.max(z) < 5.0, (np.argmax(z), np.max(z))
    assert np.all(np.abs(gm.std(axis=0) / sd - 1) < 0.25)
    assert np.mean([d['mean_accept'] for d in dg]) > 0.7


def _harness():
    so = os.path.join(ROOT, 'tests', 'host', 'liblbfgs_oracle.so')
    src = os.path.join(ROOT, 'tests', 'host', 'lbfgs_oracle.cpp')
    from oracle import oracle as orc
    orc.build()
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', src, '-L' + os.path.join(ROOT, 'oracle'),
                               '-loracle', '-Wl,-rpath,' + os.path.join(ROOT, 'oracle'), '-o', so])
    return C.CDLL(so)


def _gamma(d, x):
    tau_plot = np.logspace(-7, 2, 200)
    eps = float(d['epsilon'])
    Phi = np.exp(-(eps * np.log(tau_plot[:, None] / d['tau'][None, :])) ** 2)
    return Phi @ x


@pytest.mark.parametrize('tag,max_iter', [('K81', 3000), ('K161', 3000)])
def test_map_gpu_vs_oracle_same_optimiser(tag, max_iter):
    """BASELINE config 2: same start, same L-BFGS, GPU evaluations vs CPU-oracle evaluations: gamma(ln tau) within
    1e-4 rel-L2 (tolerance stated by north_star) after the same number of iterations."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import optimize_batch
    from oracle import oracle as orc
    blk, Z, f, kw, d = _bench_problem('optimize', tag)
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    rs = np.random.RandomState(1234)
    th0 = rs.uniform(-2, 2, prob.D)
    out, rep = optimize_batch(prob, th0[None], max_iter=max_iter)
    h = _harness()
    ref = np.empty(prob.D); it = C.c_int(); ne = C.c_int(); lp = C.c_double()
    h.harness_optimize(C.byref(om.m), th0.ctypes.data_as(C.c_void_p), max_iter, ref.ctypes.data_as(C.c_void_p),
                       C.byref(it), C.byref(ne), C.byref(lp))
    K = prob.Ks[0]
    xg, xr = np.exp(out[0][2:2 + K]), np.exp(ref[2:2 + K])
    err = rel_l2(_gamma(d, xg), _gamma(d, xr))
    assert rep[0]['lp'] > om.logp(th0, False) + 100            # it did optimise
    assert abs(rep[0]['lp'] - lp.value) < 1e-6 * abs(lp.value)
    assert err < 1e-4, (err, rep[0], it.value)


def test_batched_optimize_equals_single_fits():
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import optimize_batch
    blk, Z, f, kw, d = _bench_problem('optimize', 'K81')
    rng = np.random.default_rng(3)
    Zs = np.stack([Z, Z * 1.02 + 0.002 * rng.standard_normal(len(Z)), Z * 0.97])
    prob = Problem([blk], Zs, f, **kw)
    th0 = np.random.RandomState(5).uniform(-2, 2, (3, prob.D))
    both, rb = optimize_batch(prob, th0, spec=[0, 1, 2], max_iter=300)
    for i in range(3):
        one, r1 = optimize_batch(prob, th0[i][None], spec=[i], max_iter=300)
        assert np.array_equal(one[0], both[i]) and r1[0]['iterations'] == rb[i]['iterations']
