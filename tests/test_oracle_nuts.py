"""Oracle NUTS (oracle/nuts_oracle.c) on known-answer targets: independent Gaussians with very different scales
(exercises step-size + diagonal-metric adaptation).  Sampler validity is statistical: means within 5 standard
errors, standard deviations within 10 %."""
import numpy as np

from oracle import oracle as orc


def _ess(x):
    x = x - x.mean()
    n = len(x)
    ac = np.correlate(x, x, 'full')[n - 1:] / (x.var() * np.arange(n, 0, -1))
    s = 0.0
    for k in range(1, n // 2):
        if ac[k] < 0.05:
            break
        s += ac[k]
    return n / (1 + 2 * s)


def test_gaussian_known_answer():
    mu = np.array([0.0, 3.0, -2.0, 10.0, 0.5, 1.0])
    sd = np.array([1.0, 0.01, 100.0, 2.0, 0.3, 5.0])
    chains = [orc.nuts_sample_gauss(mu, sd, c, 99, 400, 1500) for c in range(4)]
    d = np.concatenate([c[0] for c in chains])
    diag = [c[1] for c in chains]
    assert all(x['n_divergent'] == 0 for x in diag)
    assert all(0.75 < x['mean_accept'] < 0.99 for x in diag)
    for j in range(len(mu)):
        ess = sum(_ess(c[0][:, j]) for c in chains)
        se = sd[j] / np.sqrt(ess)
        assert abs(d[:, j].mean() - mu[j]) < 5 * se, (j, d[:, j].mean(), mu[j], se)
        assert abs(d[:, j].std() / sd[j] - 1) < 0.1, (j, d[:, j].std(), sd[j])
    # metric adaptation makes trees short on an axis-aligned Gaussian
    assert np.mean([x["n_leapfrog"] for x in diag]) / (400 + 1500) < 100   # early warm-up (unit metric, scales 1e-2..1e2) needs deep trees


def test_seed_and_chain_determinism():
    mu = np.zeros(3); sd = np.array([1.0, 2.0, 0.5])
    a, _ = orc.nuts_sample_gauss(mu, sd, 0, 7, 50, 50)
    b, _ = orc.nuts_sample_gauss(mu, sd, 0, 7, 50, 50)
    c, _ = orc.nuts_sample_gauss(mu, sd, 1, 7, 50, 50)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_drt_posterior_small_runs():
    from tests.helpers import load
    d = load('dat_sample_2ZARC_uniform_0.25_K81')
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    m = orc.OracleModel([blk], d['Z'], d['freq'], ups_alpha=1.0, ups_beta=0.1)
    draws, lp, diag = orc.nuts_sample(m, 0, 1234, 8, 4, control=orc.nuts_control(max_treedepth=5))
    assert draws.shape == (4, m.D) and np.all(np.isfinite(draws)) and np.all(np.isfinite(lp))
    assert diag['n_leapfrog'] > 0
    # lp stored with the draw is the log density at the draw
    assert abs(m.logp(draws[-1], True) - lp[-1]) < 1e-9 * max(1, abs(lp[-1]))


def test_short_warmup_leaves_the_metric_alone():
    """num_warmup < 20: Stan 2.19 performs no variance estimation at all (windowed_adaptation::set_window_params returns
    early), so the metric stays at its initial value instead of being overwritten by the empty-window regulariser 1e-3
    (which made trees ~30x longer).  Known answer: a 20-d unit Gaussian, warmup=10."""
    mu = np.zeros(20); sd = np.ones(20)
    for chain in (0, 1):
        _, dw = orc.nuts_sample_gauss(mu, sd, chain, 5, 10, 0)
        _, d = orc.nuts_sample_gauss(mu, sd, chain, 5, 10, 200)
        per_draw = (d['n_leapfrog'] - dw['n_leapfrog']) / 200       # post-warm-up leapfrogs per draw
        # unit metric, adapted eps >= 0.15: half a period is at most pi / 0.15 = 21 steps => trees of <= 31 leapfrogs;
        # with the metric overwritten by 1e-3 the same eps needed ~30x as many (312 per draw measured)
        assert d['stepsize'] > 0.1 and per_draw < 40, (chain, d['stepsize'], per_draw)
