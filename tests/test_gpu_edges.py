"""GPU tests (-m gpu): edges of the evaluator's domain -- empty and ragged batches, the largest problem the fast S1 evaluator
takes (K = 192, Nf = 128), the first size beyond it (generic evaluator), a problem that does not fit the 160 KiB LDS budget
(loud error, no fallback), extreme parameter values (overflow -> non-finite lp, never a crash)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(nf, K, seed=0):
    from bayes_drt_amd import matrices as gm
    f = np.logspace(6, -2, nf)
    bf = np.logspace(8, -4, K)
    tau = 1 / (2 * np.pi * bf)
    eps = 1 / np.mean(np.diff(np.log(tau)))
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    blk = dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)
    rng = np.random.default_rng(seed)
    x = np.exp(-0.5 * ((np.log(tau) + 6) / 1.5) ** 2)
    Z = A @ x + np.concatenate([np.full(nf, 0.5), np.zeros(nf)])
    Z = Z / np.std(np.hypot(Z[:nf], Z[nf:])) + 0.01 * rng.standard_normal(2 * nf)
    kw = dict(sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0)
    return blk, Z, f, kw


def _check_vs_oracle(nf, K, n_pts, expect_fast):
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    blk, Z, f, kw = _problem(nf, K)
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    rng = np.random.default_rng(K)
    th = rng.uniform(-2, 2, (n_pts, prob.D))
    for jac in (False, True):
        lp, g = prob.logp_grad(th, jacobian=jac)
        for i in range(n_pts):
            lr, gr = om.logp_grad(th[i], jacobian=jac)
            assert abs(lp[i] - lr) <= 1e-10 * max(1.0, abs(lr))
            assert np.max(np.abs(g[i] - gr)) <= 1e-10 * max(1.0, np.max(np.abs(gr)))
    return prob


def test_largest_fast_path_problem_K192_Nf128():
    prob = _check_vs_oracle(128, 192, 21, True)
    assert prob.D == 2 * 192 + 9
    _check_vs_oracle(107, 161, 9, True)            # the length of the reference's experimental spectra (LIB: 107, PDAC: 106)


def test_first_size_beyond_the_fast_path_uses_the_generic_evaluator():
    _check_vs_oracle(81, 200, 5, False)
    _check_vs_oracle(130, 120, 5, False)          # Nf > 128: generic evaluator as well


def test_problem_beyond_the_lds_budget_takes_the_streamed_evaluator():
    """81 frequencies x 420 basis functions: refused by bdrt_problem_create until round 3 ("LDS"), evaluated by the streamed path
    of bdrt_big.h since (the reference takes any grid: inversion.py:2127-2209); the sampler's D <= 1024 holds here (D = 849)."""
    from bayes_drt_amd.model import Problem
    blk, Z, f, kw = _problem(81, 420)
    prob = Problem([blk], Z, f, **kw)
    assert prob.evaluator() == 5 and prob.D == 2 * 420 + 9
    lp, g = prob.logp_grad(np.random.default_rng(0).uniform(-1, 1, (3, prob.D)))
    assert np.all(np.isfinite(lp)) and np.all(np.isfinite(g))
    prob.close()


def test_empty_single_and_ragged_batches():
    from bayes_drt_amd.model import Problem
    blk, Z, f, kw = _problem(40, 60)
    prob = Problem([blk], Z, f, **kw)
    lp, g = prob.logp_grad(np.empty((0, prob.D)))
    assert lp.shape == (0,) and g.shape == (0, prob.D)
    rng = np.random.default_rng(3)
    th = rng.uniform(-2, 2, (33, prob.D))
    lp_all, g_all = prob.logp_grad(th)
    for B in (1, 15, 16, 17):                      # evaluations do not depend on how the batch is tiled (16 per workgroup)
        lp_b, g_b = prob.logp_grad(th[:B])
        assert np.array_equal(lp_b, lp_all[:B]) and np.array_equal(g_b, g_all[:B])


def test_extreme_parameters_give_non_finite_lp_not_a_crash():
    from bayes_drt_amd.model import Problem
    blk, Z, f, kw = _problem(81, 161)
    prob = Problem([blk], Z, f, **kw)
    th = np.zeros((4, prob.D))
    th[0, 2:50] = 800.0          # exp overflow in x
    th[1, :] = -800.0            # everything underflows
    th[2, 5] = np.nan
    lp, g = prob.logp_grad(th, jacobian=True)
    assert not np.isfinite(lp[0]) and not np.isfinite(lp[2])
    assert np.isfinite(lp[3]) and np.all(np.isfinite(g[3]))
    assert lp.shape == (4,)


@pytest.mark.timeout(180)
@pytest.mark.parametrize('units,blocks', [(4, 1), (17, 1), (4, 2), (20, 2)])
def test_a_spectrum_without_a_finite_log_posterior_is_reported_not_sampled(units, blocks):
    """Stan: "Initialization failed" after 100 attempts.  A NaN in the spectrum makes every evaluation
    non-finite: each sampler kernel (one chain per workgroup for the single DRT and for two distributions, sixteen chains per
    workgroup) must stop its chains and say so -- no hang, no draws."""
    from bayes_drt_amd._lib import BdrtError
    from bayes_drt_amd.engine import sample_units
    from bayes_drt_amd.model import Problem
    blk, Z, f, kw = _problem(40, 64)
    Z = Z.copy(); Z[7] = np.nan
    prob = Problem([blk] + [dict(blk, parallel=True, nonneg=True)] * (blocks - 1), Z, f, **kw)
    with pytest.raises(BdrtError) as e:
        sample_units(prob, units, 20, 10, 3)
    assert '100 attempts' in str(e.value)
    # the evaluator itself returns the non-finite value, it does not raise
    lp, g = prob.logp_grad(np.zeros((2, prob.D)))
    assert not np.any(np.isfinite(lp))
    prob.close()


def test_sampler_argument_checks_and_empty_runs():
    from bayes_drt_amd._lib import BdrtError
    from bayes_drt_amd.engine import sample_units
    from bayes_drt_amd.model import Problem
    blk, Z, f, kw = _problem(40, 64)
    prob = Problem([blk], Z, f, **kw)
    with pytest.raises(BdrtError):
        sample_units(prob, 0, 10, 10, 1)
    with pytest.raises(BdrtError):
        sample_units(prob, 2, -1, 10, 1)
    with pytest.raises(BdrtError):
        sample_units(prob, 2, 10, 10, 1, spec=[0, 1])            # spectrum 1 does not exist
    import ctypes as C
    from bayes_drt_amd._lib import NutsControl
    for bad in (dict(adapt_delta=1.0), dict(adapt_delta=0.0), dict(adapt_delta=float('nan')), dict(stepsize0=0.0), dict(stepsize0=float('inf')),
                dict(stepsize0=-1.0), dict(adapt_gamma=0.0), dict(adapt_t0=-5.0), dict(max_deltaH=float('nan')), dict(init_radius=-1.0),
                dict(base_window=-1), dict(max_treedepth=0), dict(max_treedepth=11)):
        c = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(c))
        for k, v in bad.items():
            setattr(c, k, v)
        with pytest.raises(BdrtError):                                # Stan's own argument checks; NaN fails them too
            sample_units(prob, 2, 10, 5, 1, c)
    draws, lp, diag = sample_units(prob, 3, 12, 0, 1)             # warm-up only: no draws, the chains still ran
    assert draws.shape == (3, 0, prob.D) and all(d['n_leapfrog'] > 0 for d in diag)
    draws, lp, diag = sample_units(prob, 3, 0, 5, 1)              # no warm-up: unit step size search, then draws
    assert draws.shape == (3, 5, prob.D) and np.all(np.isfinite(draws))
    prob.close()


@pytest.mark.timeout(120)
def test_non_finite_or_degenerate_spectra_are_refused_not_hung_on():
    """NaN / inf in Z, non-positive frequencies, a spectrum without spread: `Inverter` refuses them before any kernel runs (they
    used to spin in the KKT regularisation loop of the QP kernel -- on the device -- because `reg > 1e6` is false for NaN).  The
    C ABI itself returns an error for a non-finite QP and a non-zero return code for a MAP without a finite log-posterior."""
    import warnings
    from bayes_drt_amd._lib import BdrtError
    from bayes_drt_amd.engine import optimize_batch
    from bayes_drt_amd.inversion import Inverter, _qp_batch
    from bayes_drt_amd.model import Problem
    f = np.logspace(5, -1, 31)
    w = 2 * np.pi * f
    Z = 1.0 + 2.0 / (1 + (1j * w * 1e-2) ** 0.9)
    for bad in (np.nan, np.inf):
        Zb = Z.copy(); Zb[4] = bad
        for call in (lambda inv: inv.fit(f, Zb, nonneg=True), lambda inv: inv.ridge_fit(f, Zb),
                     lambda inv: inv.fit(f, Zb, nonneg=True, mode='sample', warmup=5, samples=2),
                     lambda inv: inv.ridge_ReImCV(f, Zb, lambdas=np.logspace(-3, 0, 3))):
            with pytest.raises(ValueError):
                call(Inverter(basis_freq=f))
    with pytest.raises(ValueError):
        Inverter(basis_freq=f).fit(np.r_[f[:-1], 0.0], Z, nonneg=True)
    with pytest.raises(ValueError):
        Inverter(basis_freq=f).ridge_fit(f, np.zeros_like(Z))
    # the C ABI on its own
    n = 12
    rng = np.random.default_rng(0)
    A = rng.standard_normal((20, n)); P = A.T @ A; q = rng.standard_normal(n)
    Pn = P.copy(); Pn[3, 3] = np.nan
    with pytest.raises(BdrtError):
        _qp_batch(Pn[None], q[None], np.zeros(n))
    x, obj = _qp_batch(P[None], q[None], np.zeros(n))                  # (and the kernel still works afterwards)
    assert np.all(np.isfinite(x))
    blk, Zs, fs, kw = _problem(40, 64)
    for key in ('A', 'L1'):                                            # matrices must be finite: refused at creation
        bad = dict(blk); bad[key] = blk[key].copy(); bad[key][3, 5] = np.inf if key == 'L1' else np.nan
        with pytest.raises(BdrtError) as e:
            Problem([bad], Zs, fs, **kw)
        assert 'non-finite' in str(e.value)
    from bayes_drt_amd import post
    with pytest.raises(ValueError):
        post.percentile(np.ones((5, 3)), np.nan)                       # numpy: "Percentiles must be in the range [0, 100]"
    Zs = Zs.copy(); Zs[3] = np.nan
    prob = Problem([blk], Zs, fs, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out, rep = optimize_batch(prob, np.zeros((1, prob.D)))
    assert rep[0]['return_code'] != 0
    prob.close()


def test_device_memory_does_not_grow_over_many_problems_samplers_and_fits():
    """Create / use / destroy cycles (problems, MAP, samplers of the three kernel kinds, Inverter ridge + MAP + NUTS fits): the
    free device memory after 40 more cycles equals the free memory after the first ones."""
    import gc
    import warnings
    import torch
    from bayes_drt_amd.engine import optimize_batch, sample_units
    from bayes_drt_amd.inversion import Inverter
    from bayes_drt_amd.model import Problem

    def free_mib():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0] / 2 ** 20

    blk, Z, f, kw = _problem(81, 161)
    fz = np.logspace(5, -1, 41)
    Zz = 1.0 + 2.0 / (1 + (2j * np.pi * fz * 1e-2) ** 0.9)

    def cycle(i):
        prob = Problem([blk], Z, f, **kw)
        prob.logp_grad(np.zeros((5, prob.D)))
        optimize_batch(prob, np.zeros((1, prob.D)), newton_max_iter=3)
        sample_units(prob, [3, 20, 300][i % 3], 5, 3, i)
        prob.close()
        inv = Inverter(basis_freq=fz)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            inv.ridge_fit(fz, Zz)
            inv.fit(fz, Zz, nonneg=True)
            if i % 10 == 0:
                inv.fit(fz, Zz, nonneg=True, mode='sample', warmup=10, samples=5)
        inv.predict_Z(fz)

    for i in range(6):
        cycle(i)
    gc.collect()
    before = free_mib()
    for i in range(40):
        cycle(i)
    gc.collect()
    after = free_mib()
    assert before - after < 8.0, (before, after)


@pytest.mark.timeout(300)
def test_concurrent_host_threads_get_the_sequential_results():
    """Six host threads, each with its own problem, MAP and sampler (all three kernel kinds), twelve rounds: every result is
    bit-identical to the sequential run.  (It used to fail about once in twenty rounds: `hipMemset` of the draws buffer and the
    leapfrog counter returns before the fill happens, and the sampler's non-blocking stream did not wait for it.)"""
    import threading
    from bayes_drt_amd.engine import optimize_batch, sample_units
    from bayes_drt_amd.model import Problem

    def job(i):
        blk, Z, f, kw = _problem(40 + 8 * (i % 3), 64 + 16 * (i % 2), seed=i)
        prob = Problem([blk], Z, f, **kw)
        lp, g = prob.logp_grad(np.full((3, prob.D), 0.1 * i))
        out, rep = optimize_batch(prob, np.zeros((1, prob.D)))
        d, l, dg = sample_units(prob, [3, 20, 40][i % 3], 10, 5, 100 + i)
        prob.close()
        return lp, out, np.float64(rep[0]['lp']), d, np.array([x['n_leapfrog'] for x in dg])

    n = 6
    seq = [job(i) for i in range(n)]
    for rnd in range(12):
        res, err = [None] * n, []

        def run(i):
            try:
                res[i] = job(i)
            except Exception as e:           # noqa: BLE001 -- reported below
                err.append((i, repr(e)))
        th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not err, err
        for i in range(n):
            for a, b in zip(seq[i], res[i]):
                assert np.array_equal(a, b), (rnd, i)
