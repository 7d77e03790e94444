"""CPU oracle leg of the RC_Macdonald_0.25 study: the recursive NUTS of oracle/nuts_oracle.c on the spectrum of the
published HMC study whose stored posterior mean sits 17.5 % from ours, with the notebook's model (basis = measurement frequencies,
K = 81, Series_pos, sigma_min 0.002) -- 4 chains x (warmup + draws), one process per chain.  Writes tests/golden/rc_macdonald_oracle.npz:
the posterior mean / 2.5 % / 97.5 % of the constrained coefficients and offsets over all chains, per-chain means and diagnostics.
The Stan data dict comes from the Inverter's host code with the oracle's matrix builders in place of the GPU's (as tests/test_inverter_host.py).
Usage: python tools/rc_macdonald_oracle.py [warmup draws]    (CPU only; ~10 min on four cores at 1000 + 1000)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiprocessing import Pool
from oracle import oracle as orc
from tests.helpers import load

STEM = 'RC_Macdonald_0.25'


def stan_data():
    from bayes_drt_amd import inversion
    inversion.construct_A = lambda frequencies, part, tau=None, basis='gaussian', fit_inductance=False, epsilon=1, kernel='DRT', dist_type='series', symmetry='planar', bc=None, ct=False, k_ct=None, integrate_method='trapz': \
        orc.construct_A(frequencies, part, tau=tau, epsilon=epsilon, kernel=kernel, dist_type=dist_type, symmetry=symmetry, bc=bc if bc else None, ct=ct, k_ct=k_ct)
    inversion.construct_L = lambda frequencies, tau=None, basis='gaussian', epsilon=1, order=1: orc.construct_L(tau, epsilon, order)
    S = load('hmc_suite')
    i = [str(s) for s in S['stems']].index(STEM)
    f, Z = S['Z'][i][:, 0], S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2]
    inv = inversion.Inverter(basis_freq=f)
    fs, Zs, _, _, _, _, dm = inv._prep_matrices(f, Z, 'both', weights=None, dZ=False, scale_Z=True, penalty='discrete', fit_type='map')
    dat = inv._prep_stan_data(fs, Zs, 'both', 'Series', dm, False, 0.002, mode='sample', inductance_scale=1, outlier_lambda=None,
                              fitY=False, SA=False, SASY=False)
    return dat, float(inv._Z_scale)


def chain_run(args):
    chain, warm, draws = args
    dat, _ = stan_data()
    blk = dict(A=dat['A'], L0=dat['L0'], L1=dat['L1'], L2=dat['L2'], nonneg=True)
    m = orc.OracleModel([blk], dat['Z'], dat['freq'], sigma_min=float(dat['sigma_min']), ups_alpha=float(dat['ups_alpha']),
                        ups_beta=float(dat['ups_beta']))
    t0 = time.time()
    th, lp, d = orc.nuts_sample(m, chain, 1234, warm, draws, control=orc.nuts_control(adapt_delta=0.9, adapt_t0=10.0))
    return chain, th, lp, d, time.time() - t0


if __name__ == '__main__':
    warm, draws = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 1000)
    os.environ.setdefault('BDRT_ORACLE_NATIVE', '1')
    dat, zscale = stan_data()
    K = dat['A'].shape[1]
    with Pool(4) as pool:
        res = sorted(pool.map(chain_run, [(c, warm, draws) for c in range(4)]))
    D = res[0][1].shape[1]
    theta = np.concatenate([r[1] for r in res])                        # [4 * draws, D] unconstrained
    # Series_pos layout (include/bdrt.h): Rinf_raw, induc_raw, x[K] (log), sigma_res.., ups[K] (log), d[3] (log)
    x = np.exp(theta[:, 2:2 + K]) * zscale
    rinf = 100.0 * np.exp(theta[:, 0]) * zscale                       # Rinf_raw is declared <lower=0>
    out = dict(stem=STEM, warmup=warm, draws=draws, seed=1234, K=K, x_mean=x.mean(axis=0), x_lo=np.percentile(x, 2.5, axis=0),
               x_hi=np.percentile(x, 97.5, axis=0), Rinf_mean=rinf.mean(), lp_mean=np.mean([r[2].mean() for r in res]),
               x_chain_mean=np.stack([np.exp(r[1][:, 2:2 + K]).mean(axis=0) * zscale for r in res]),
               n_leapfrog=np.array([r[3]['n_leapfrog'] for r in res]), stepsize=np.array([r[3]['stepsize'] for r in res]),
               n_max_treedepth=np.array([r[3]['n_max_treedepth'] for r in res]), n_divergent=np.array([r[3]['n_divergent'] for r in res]),
               seconds=np.array([r[4] for r in res]))
    for k in ('n_leapfrog', 'stepsize', 'n_max_treedepth', 'n_divergent', 'seconds'):
        print(k, out[k])
    cm = out['x_chain_mean']
    print('chain-to-chain spread of the coefficient means (rel-L2 against the pooled mean):',
          [float(np.linalg.norm(c - out['x_mean']) / np.linalg.norm(out['x_mean'])) for c in cm])
    if warm >= 500:
        np.savez(os.path.join(ROOT, 'tests', 'golden', 'rc_macdonald_oracle.npz'), **out)
        print('written tests/golden/rc_macdonald_oracle.npz')
