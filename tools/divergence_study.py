"""Divergent iterations of the published HMC study: is "26 ... 142 of 24 000 against pystan's 15" the
scatter of start points, or a systematic offset of the adapted step size?

Per spectrum of tests/golden/hmc_suite.npz (the 60 simulated spectra of code_EchemActa/Run fits.ipynb cell 5, 2 chains x (200 + 200),
adapt_delta 0.9, adapt_t0 10: bayes_drt/inversion.py:1218-1221) and per seed: the adapted step size, the mean acceptance statistic, the
divergent and the depth-saturated post-warm-up iterations of every chain --
  hip      the product: ONE Inverter.fit_many call per seed for all 60 spectra (per-spectrum nonneg / sigma_min lists)      [GPU box]
  oracle   oracle/nuts_oracle.c, one process per (seed, spectrum, chain); same Philox streams, hence the same start points    [CPU]
  outliers the same table for the outlier error model (Series_pos_outliers, 100 spectra x 4 chains: tools/wave_outliers_study.py's job) [GPU box]
  report   both records against pystan's printed per-spectrum counts (hmc_suite.npz['diag'])
Usage: divergence_study.py hip|oracle|outliers --seeds 1234,1,2,... [--out file.npz] [--procs n] [--stems a,b]
       divergence_study.py report hip.npz [oracle.npz]"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import load, rel_l2      # noqa: E402

TAU_PLOT = np.logspace(-7, 2, 200)
COLS = ('stepsize', 'mean_accept', 'n_divergent', 'n_max_treedepth', 'n_leapfrog')


def _arg(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def _study():
    S = load('hmc_suite')
    stems = [str(s) for s in S['stems']]
    f = S['Z'][0][:, 0]
    Z = [S['Z'][i][:, 1] + 1j * S['Z'][i][:, 2] for i in range(len(stems))]
    nonneg = [not s.startswith('ZARC-RL') for s in stems]
    smin = [0.005 if 'noiseless' in s else 0.002 for s in stems]
    return S, stems, f, Z, nonneg, smin


def run_hip(seeds, out):
    from bayes_drt_amd.inversion import Inverter
    S, stems, f, Z, nonneg, smin = _study()
    rec = np.zeros((len(seeds), len(stems), 2, len(COLS)))
    err = np.zeros((len(seeds), len(stems), 3))
    for a, seed in enumerate(seeds):
        t0 = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            views = Inverter(basis_freq=f).fit_many(f, Z, nonneg=nonneg, sigma_min=smin, mode='sample', warmup=200, samples=200, chains=2,
                                                    random_seed=seed)
        for i, v in enumerate(views):
            for c, d in enumerate(v._sample_result.diagnostics):
                rec[a, i, c] = [d[k] for k in COLS]
            ref = S['Gout_bayes'][i]
            err[a, i] = [rel_l2(v.predict_distribution('DRT', eval_tau=TAU_PLOT), ref[:, 1])] + \
                        [rel_l2(v.predict_distribution('DRT', eval_tau=TAU_PLOT, percentile=p), ref[:, k]) for p, k in ((2.5, 2), (97.5, 3))]
        print('# hip seed %d: %.1f s; divergent %d, saturated %d of 24000; mean-curve error median %.4f, spectra above 4 %%: %d' % (
            seed, time.time() - t0, rec[a, :, :, 2].sum(), rec[a, :, :, 3].sum(), np.median(err[a, :, 0]), int((err[a, :, 0] > 0.04).sum())), flush=True)
    np.savez(out, seeds=np.array(seeds), stems=np.array(stems), rec=rec, err=err, cols=np.array(COLS), leg='hip')


_ORC = {}


def _oracle_dat(i):
    """Stan data of spectrum i from the Inverter's host code with the oracle's matrix builders in place of the GPU's."""
    if i in _ORC:
        return _ORC[i]
    from bayes_drt_amd import inversion
    from oracle import oracle as orc
    inversion.construct_A = lambda frequencies, part, tau=None, basis='gaussian', fit_inductance=False, epsilon=1, kernel='DRT', dist_type='series', symmetry='planar', bc=None, ct=False, k_ct=None, integrate_method='trapz': \
        orc.construct_A(frequencies, part, tau=tau, epsilon=epsilon, kernel=kernel, dist_type=dist_type, symmetry=symmetry, bc=bc if bc else None, ct=ct, k_ct=k_ct)
    inversion.construct_L = lambda frequencies, tau=None, basis='gaussian', epsilon=1, order=1: orc.construct_L(tau, epsilon, order)
    S, stems, f, Z, nonneg, smin = _study()
    inv = inversion.Inverter(basis_freq=f)
    fs, Zs, _, _, _, _, dm = inv._prep_matrices(f, Z[i], 'both', weights=None, dZ=False, scale_Z=True, penalty='discrete', fit_type='map')
    dat = inv._prep_stan_data(fs, Zs, 'both', 'Series', dm, False, smin[i], mode='sample', inductance_scale=1, outlier_lambda=None,
                              fitY=False, SA=False, SASY=False)
    _ORC[i] = (dat, nonneg[i])
    return _ORC[i]


def _oracle_chain(args):
    seed, i, chain = args
    from oracle import oracle as orc
    dat, nonneg = _oracle_dat(i)
    blk = dict(A=dat['A'], L0=dat['L0'], L1=dat['L1'], L2=dat['L2'], nonneg=nonneg)
    m = orc.OracleModel([blk], dat['Z'], dat['freq'], sigma_min=float(dat['sigma_min']), ups_alpha=float(dat['ups_alpha']),
                        ups_beta=float(dat['ups_beta']))
    _, _, d = orc.nuts_sample(m, chain, seed, 200, 200, control=orc.nuts_control(adapt_delta=0.9, adapt_t0=10.0))
    return seed, i, chain, [d[k] for k in COLS]


def run_oracle(seeds, out, procs, only):
    from multiprocessing import Pool
    os.environ.setdefault('BDRT_ORACLE_NATIVE', '1')
    S, stems, f, Z, nonneg, smin = _study()
    idx = [i for i, s in enumerate(stems) if only is None or s in only]
    rec = np.full((len(seeds), len(stems), 2, len(COLS)), np.nan)
    jobs = [(seed, i, c) for seed in seeds for i in idx for c in (0, 1)]
    t0 = time.time()
    with Pool(procs) as pool:
        for n, (seed, i, c, row) in enumerate(pool.imap_unordered(_oracle_chain, jobs)):
            rec[seeds.index(seed), i, c] = row
            if (n + 1) % 60 == 0:
                print('# oracle: %d of %d chains, %.0f s' % (n + 1, len(jobs), time.time() - t0), flush=True)
                np.savez(out, seeds=np.array(seeds), stems=np.array(stems), rec=rec, cols=np.array(COLS), leg='oracle')
    np.savez(out, seeds=np.array(seeds), stems=np.array(stems), rec=rec, cols=np.array(COLS), leg='oracle')
    print('# oracle: %d chains in %.0f s on %d processes' % (len(jobs), time.time() - t0, procs))


def run_outliers(seeds, out):
    """The outlier error model in the shape tools/wave_outliers_study.py uses it: 100 spectra x 4 chains x (150 + 150)."""
    from bayes_drt_amd.inversion import Inverter
    S, stems, f, Z, nonneg, smin = _study()
    rs = np.random.RandomState(3)
    pick = [i for i in range(len(stems)) if nonneg[i]][:50]
    Zs = []
    for rep in range(2):
        for i in pick:
            z = Z[i].copy()
            for j in rs.choice(len(z), 3, replace=False):
                z[j] *= 1.5
            Zs.append(z)
    rec = np.zeros((len(seeds), len(Zs), 4, len(COLS)))
    for a, seed in enumerate(seeds):
        t0 = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            views = Inverter(basis_freq=f).fit_many(f, Zs, nonneg=True, outliers=True, mode='sample', warmup=150, samples=150, chains=4,
                                                    random_seed=seed)
        for i, v in enumerate(views):
            for c, d in enumerate(v._sample_result.diagnostics):
                rec[a, i, c] = [d[k] for k in COLS]
        r = rec[a]
        print('# outlier model seed %d: %.1f s; divergent %d of %d post-warm-up iterations (%.2f %%), saturated %d; chains with > 10 divergent: %d of %d; '
              'step size median %.4f, mean accept median %.3f' % (seed, time.time() - t0, r[:, :, 2].sum(), 150 * r.shape[0] * 4,
                                                                  100 * r[:, :, 2].sum() / (150 * r.shape[0] * 4), r[:, :, 3].sum(),
                                                                  int((r[:, :, 2] > 10).sum()), r.shape[0] * 4, np.median(r[:, :, 0]),
                                                                  np.median(r[:, :, 1])), flush=True)
    np.savez(out, seeds=np.array(seeds), rec=rec, cols=np.array(COLS), leg='outliers')


def report(paths):
    S, stems, f, Z, nonneg, smin = _study()
    ref = S['diag']
    legs = [np.load(p) for p in paths]
    for L in legs:
        rec, seeds = L['rec'], [int(s) for s in L['seeds']]
        leg = str(L['leg'])
        have = ~np.isnan(rec[:, :, 0, 0])
        print('=== %s: %d seeds %s' % (leg, len(seeds), seeds))
        print('%-26s | pystan sat div | per seed: divergent (both chains) | saturated | step sizes of the two chains (median over seeds) | accept (median)' % 'spectrum')
        for i, s in enumerate(stems):
            if not have[:, i].any():
                continue
            r = rec[have[:, i], i]
            div = r[:, :, 2].sum(axis=1).astype(int); sat = r[:, :, 3].sum(axis=1).astype(int)
            print('%-26s | %4d %3d | %-44s | %-50s | %.4f %.4f | %.3f' % (s, ref[i, 0], ref[i, 1], ' '.join('%d' % d for d in div), ' '.join('%d' % d for d in sat),
                                                                       np.median(r[:, 0, 0]), np.median(r[:, 1, 0]), np.median(r[:, :, 1])))
        tot_div = np.array([np.nansum(rec[a, :, :, 2]) for a in range(len(seeds))]); tot_sat = np.array([np.nansum(rec[a, :, :, 3]) for a in range(len(seeds))])
        print('totals per seed: divergent %s (pystan: %d); saturated %s (pystan: %d)' % (tot_div.astype(int).tolist(), ref[:, 1].sum(), tot_sat.astype(int).tolist(), ref[:, 0].sum()))
        print('divergent per study: min %d, quartiles %d / %d / %d, max %d; studies with <= 15: %d of %d' % (
            tot_div.min(), np.percentile(tot_div, 25), np.median(tot_div), np.percentile(tot_div, 75), tot_div.max(), int((tot_div <= 15).sum()), len(seeds)))
        # where the divergent iterations sit: chains with many of them
        ch = rec[:, :, :, 2][~np.isnan(rec[:, :, :, 2])]
        print('chains: %d in all; without divergent iterations %d; 1-2: %d; 3-10: %d; > 10: %d (these hold %d of the %d divergent iterations)' % (
            ch.size, int((ch == 0).sum()), int(((ch >= 1) & (ch <= 2)).sum()), int(((ch >= 3) & (ch <= 10)).sum()), int((ch > 10).sum()),
            int(ch[ch > 10].sum()), int(ch.sum())))
        # spectra on which pystan reported divergent iterations vs ours
        pos = ref[:, 1] > 0
        per = np.nansum(rec[:, :, :, 2], axis=2)          # [seed, spectrum]
        print('spectra with divergent iterations in pystan (%d spectra, %d iterations): ours per seed on THOSE spectra median %.1f; on the other %d spectra median %.1f' % (
            int(pos.sum()), int(ref[pos, 1].sum()), np.median(per[:, pos].sum(axis=1)), int((~pos).sum()), np.median(per[:, ~pos].sum(axis=1))))
        if 'err' in L.files:
            e = L['err']
            med = np.median(e, axis=0)
            print('posterior-mean curve vs stored, per-spectrum MEDIAN over seeds: median %.4f, > 4 %%: %s' % (
                np.median(med[:, 0]), [(stems[i], round(float(med[i, 0]), 4)) for i in np.nonzero(med[:, 0] > 0.04)[0]]))
            print('97.5 %% curve vs stored, per-spectrum MEDIAN over seeds: median %.4f, > 10 %%: %s' % (
                np.median(med[:, 2]), [(stems[i], round(float(med[i, 2]), 4)) for i in np.nonzero(med[:, 2] > 0.10)[0]]))
    if len(legs) == 2:
        a, b = legs
        sa, sb = [int(s) for s in a['seeds']], [int(s) for s in b['seeds']]
        common = [s for s in sa if s in sb]
        print('=== hip vs oracle on the same (seed, spectrum, chain): %d common seeds %s' % (len(common), common))
        ra = np.stack([a['rec'][sa.index(s)] for s in common]); rb = np.stack([b['rec'][sb.index(s)] for s in common])
        ok = ~np.isnan(rb[..., 0])
        ea, eb = ra[..., 0][ok], rb[..., 0][ok]
        ratio = ea / eb
        print('adapted step size hip / oracle over %d chains: median %.4f, quartiles %.3f / %.3f, within 1 %%: %d, within 10 %%: %d' % (
            ratio.size, np.median(ratio), np.percentile(ratio, 25), np.percentile(ratio, 75), int((np.abs(ratio - 1) < 0.01).sum()), int((np.abs(ratio - 1) < 0.1).sum())))
        print('log step size: mean hip %.4f oracle %.4f (difference %.4f +- %.4f)' % (np.mean(np.log(ea)), np.mean(np.log(eb)), np.mean(np.log(ratio)),
                                                                                     np.std(np.log(ratio)) / np.sqrt(ratio.size)))
        print('mean accept: hip %.4f oracle %.4f; divergent iterations: hip %d oracle %d; saturated: hip %d oracle %d' % (
            ra[..., 1][ok].mean(), rb[..., 1][ok].mean(), ra[..., 2][ok].sum(), rb[..., 2][ok].sum(), ra[..., 3][ok].sum(), rb[..., 3][ok].sum()))


if __name__ == '__main__':
    mode = sys.argv[1]
    if mode == 'report':
        report(sys.argv[2:])
    else:
        seeds = [int(s) for s in _arg('--seeds', '1234').split(',')]
        out = _arg('--out', os.path.join(ROOT, 'gpurun_out', 'divergence_%s.npz' % mode))
        os.makedirs(os.path.dirname(out), exist_ok=True)
        if mode == 'hip':
            run_hip(seeds, out)
        elif mode == 'oracle':
            only = _arg('--stems')
            run_oracle(seeds, out, int(_arg('--procs', '6')), set(only.split(',')) if only else None)
        elif mode == 'outliers':
            run_outliers(seeds, out)
        else:
            raise SystemExit(__doc__)
