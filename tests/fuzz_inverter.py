"""Randomised end-to-end cases of `Inverter.fit` (the reference's entry point, inversion.py:1072): random two-ZARC spectra
(16 ... 128 frequencies, noise 0 ... 1 %, optional outliers), random options (nonneg, outliers False / True / 'auto',
init_from_ridge, sigma_min, inductance_scale, default or extended basis), MAP mostly, short NUTS runs sometimes; one case
in six adds a finite-length diffusion element and is fitted with a DRT + a parallel transmissive DDT (Series-Parallel models).

No reference output exists for random inputs, so the checks are properties every correct fit has:
  1. no exception; every reported quantity finite; the MAP either converged (|grad|_inf < 1e-8) or says why not;
  2. the fitted impedance follows the spectrum: rms residual <= 5 noise levels + 3 % of mean |Z| (outlier points excluded;
     5 % / 15 % for 21- / 16-point spectra -- 1 to 2 points per decade, below what the model was calibrated for -- and + 5 %
     when outliers are present and the error model was told to ignore them);
  3. a non-negative fit has gamma >= 0 everywhere; R_inf and the polarisation resistance are within 25 % (+ noise) of the
     spectrum's true values (60 % for 16-point spectra and for the short NUTS runs) -- the relaxations lie two decades inside
     the frequency window;
  4. NUTS (2 chains x (60 + 40), far from converged): the percentile band brackets its own median, the posterior-mean
     impedance stays within 40 % of mean |Z| of the spectrum (a sanity band);
  5. every prediction entry point (predict_Z on fitted and new frequencies, predict_distribution, predict_Rp, predict_sigma,
     coef_percentile, score; with percentiles for NUTS fits) returns finite, ordered numbers, and a fresh Inverter that loads the
     saved fit predicts bit-identical values.

Test infrastructure: `python -m tests.fuzz_inverter --count 100` (record: profiles/r02/fuzz_inverter.txt)."""
import argparse
import sys
import time
import warnings

import numpy as np


def make_case(n):
    rng = np.random.default_rng(104729 * n + 11)
    nf = int(rng.choice([16, 21, 41, 61, 81, 101, 128]))        # (10 points over 6-9 decades: no DRT model has the capacity)
    hi, lo = float(rng.uniform(4.5, 7)), float(rng.uniform(-3, -0.5))
    f = np.logspace(hi, lo, nf)
    w = 2 * np.pi * f
    R0 = float(rng.uniform(0.2, 2.0))
    Z = np.full(nf, R0, dtype=complex)
    Rp = 0.0
    for _ in range(2):
        R = float(rng.uniform(0.5, 3.0))
        t0 = float(10 ** rng.uniform(-hi + 2.0, -lo - 2.0))          # relaxations inside the window
        al = float(rng.uniform(0.8, 1.0))
        Z = Z + R / (1 + (1j * w * t0) ** al)
        Rp += R
    # one case in six: a finite-length (transmissive, planar) diffusion element in series with ONE of the arcs, fitted with a
    # DRT + a parallel DDT (the Series-Parallel models; the element is a single line of the DDT: Y = sqrt(s) coth(sqrt(s)) / R_d)
    rng2 = np.random.default_rng(15485863 * n + 5)           # (its own stream: the single-DRT cases keep their numbers)
    multi = bool(rng2.random() < 1 / 6) and nf >= 41
    if multi:
        Rd, td = float(rng2.uniform(0.5, 2.0)), float(10 ** rng2.uniform(-lo - 2.5, -lo - 1.0) / (2 * np.pi))
        sq = np.sqrt(1j * w * td)
        Z = Z + Rd * np.tanh(sq) / sq
        Rp += Rd
    noise = float(rng.choice([0.0, 0.0025, 0.01]))
    sig = noise * np.mean(np.abs(Z))
    Z = Z + sig * (rng.standard_normal(nf) + 1j * rng.standard_normal(nf))
    out_idx = []
    if rng.random() < 0.3 and nf >= 41:
        out_idx = sorted(int(i) for i in rng.choice(nf, 2, replace=False))
        for i in out_idx:
            Z[i] *= 1.4
    ext = rng.random() < 0.4
    if ext and nf <= 101:
        ppd = (nf - 1) / (hi - lo)
        n_ext = int(min(int(ppd), (161 - nf) // 2))
        bf = np.logspace(hi + n_ext / ppd, lo - n_ext / ppd, nf + 2 * n_ext)
    else:
        bf = f if rng.random() < 0.5 else None
    kw = dict(nonneg=bool(rng.random() < 0.75), outliers=(rng.choice([False, True, 'auto']) if out_idx else rng.choice([False, False, 'auto'])),
              init_from_ridge=bool(rng.random() < 0.3), sigma_min=float(rng.choice([0.002, 0.005])),
              inductance_scale=float(rng.choice([1.0, 0.1])))
    kw['outliers'] = {'False': False, 'True': True, 'auto': 'auto'}[str(kw['outliers'])]
    mode = 'sample' if rng.random() < 0.12 else 'optimize'
    if multi:
        kw.update(nonneg=True, init_from_ridge=False)            # (ridge initialisation is defined for one distribution only)
        # (part='real' / 'imag' is not drawn: the reference zeroes the other part's rows of A_p as well, so the parallel branch
        #  becomes 1 / Re Y resp. 1 / Im Y -- a different model whose fit quality is not a property one can test; the Stan data
        #  of those fits is compared with the reference's in tests/test_inverter_host.py)
        if kw['outliers'] == 'auto':
            kw['outliers'] = False                               # (so is the ridge-based outlier screening)
    if mode == 'sample':
        kw.update(mode='sample', warmup=60, samples=40, chains=2, random_seed=int(rng.integers(1, 10 ** 5)))
    text = 'nf=%d noise=%.4f outl_pts=%s basis=%s %s' % (nf, noise, out_idx, 'default' if bf is None else ('f' if len(bf) == nf else 'K=%d' % len(bf)),
                                                        ' '.join('%s=%s' % kv for kv in kw.items()))
    if rng2.random() < 0.2:
        # the caller hands the spectrum over in ascending frequency order (the reference sorts it itself, :2139-2141)
        # (the basis stays in the reference's convention, descending: it is not sorted by either implementation, and an ascending
        #  basis flips the sign of the ln-tau integrals in both)
        if bf is f:
            bf = f.copy()
        f, Z, out_idx = f[::-1].copy(), Z[::-1].copy(), sorted(nf - 1 - i for i in out_idx)
        text = 'ascending-f ' + text
    if multi:
        text = 'DRT+TP-DDT ' + text
    return dict(f=f, Z=Z, bf=bf, kw=kw, sig=sig, R0=R0, Rp=Rp, out_idx=out_idx, mode=mode, multi=multi), text


def _api_sweep(inv, case, tau):
    """Every prediction entry point of the fitted object, with and without percentiles, then a save / load round trip into a
    fresh Inverter that must predict the same numbers (reference :3980-4064)."""
    from bayes_drt_amd.inversion import Inverter
    f = case['f']
    fails = []
    f_new = np.logspace(np.log10(f[0]) - 0.3, np.log10(f[-1]) + 0.3, 37)       # frequencies that were not fitted
    sample = case['mode'] == 'sample'
    qs = (None, 2.5, 50, 97.5) if sample else (None,)
    name = 'DRT'
    out = {}
    try:
        for q in qs:
            out['Z', q] = inv.predict_Z(f_new, percentile=q)
            out['Zf', q] = inv.predict_Z(f, percentile=q)
            out['g', q] = inv.predict_distribution(name, eval_tau=tau, percentile=q)
            out['Rp', q] = np.atleast_1d(inv.predict_Rp(percentile=q))
            s_re, s_im = inv.predict_sigma(f, percentile=q)
            out['sre', q], out['sim', q] = s_re, s_im
            if sample and q is not None:
                out['coef', q] = inv.coef_percentile(name, q)
        out['score'] = np.atleast_1d(inv.score(f, case['Z']))
    except Exception as e:
        import traceback
        return ['prediction API raised %s: %s | %s' % (type(e).__name__, e, traceback.format_exc()[-300:].replace('\n', ' / '))]
    for k, v in out.items():
        v = np.asarray(v)
        if not np.all(np.isfinite(v.real if np.iscomplexobj(v) else v)):
            fails.append('%s not finite' % (k,))
    if np.any(out['sre', None] <= 0) or np.any(out['sim', None] <= 0):
        fails.append('predict_sigma not positive')
    if sample:
        # (predict_sigma on frequencies that are not exactly f_train -- e.g. in another order -- is rebuilt from percentiles of
        #  the error-model parameters, reference :3096-3139: not monotone in the percentile by construction)
        same_grid = len(f) == len(inv.f_train) and np.array_equal(np.asarray(f), np.asarray(inv.f_train))
        for key in ('g', 'Rp') + (('sre', 'sim') if same_grid else ()):
            if not (np.all(out[key, 2.5] <= out[key, 50] + 1e-12) and np.all(out[key, 50] <= out[key, 97.5] + 1e-12)):
                fails.append('%s percentiles not ordered' % key)
    # persistence
    try:
        d = inv.save_fit_data(which='core')
        new = Inverter(basis_freq=case['bf'], distributions=inv.distributions) if case['multi'] else Inverter(basis_freq=case['bf'])
        new.load_fit_data(d)
        for q in qs:
            if not np.array_equal(new.predict_Z(f_new, percentile=q), out['Z', q]):
                fails.append('after load: predict_Z(percentile=%s) differs' % q)
            if not np.array_equal(new.predict_distribution(name, eval_tau=tau, percentile=q), out['g', q]):
                fails.append('after load: predict_distribution(percentile=%s) differs' % q)
    except Exception as e:
        fails.append('save / load raised %s: %s' % (type(e).__name__, e))
    return fails


def run_case(n):
    from bayes_drt_amd.inversion import Inverter
    case, text = make_case(n)
    f, Z, kw = case['f'], case['Z'], case['kw']
    fails = []
    if case['multi']:
        inv = Inverter(basis_freq=case['bf'], distributions={
            'DRT': {'kernel': 'DRT'},
            'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel'}})
    else:
        inv = Inverter(basis_freq=case['bf'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        inv.fit(f, Z, **kw)
    info = inv.stan_model_name.replace('_StanModel.pkl', '')
    if case['mode'] == 'optimize':
        rep = inv._opt_report
        conv = rep['return_code'] == 0 and rep['grad_inf'] < 1e-8
        info += ' map=%s' % ('converged/%d' % rep['newton_iterations'] if conv else 'rc%d,|g|=%.1e' % (rep['return_code'], rep['grad_inf']))
        if len(rep.get('starts', ())) > 1:
            others = [x['lp'] for i, x in enumerate(rep['starts']) if i != rep['start'] and x['return_code'] == 0]
            if others:
                info += ' start=%d(lp %+.1f vs the best other start)' % (rep['start'], rep['lp'] - max(others))
        if not conv and rep['return_code'] not in (1, 2):
            fails.append('MAP report %r' % (rep,))
    Zp = inv.predict_Z(f)
    keep = np.ones(len(f), bool); keep[case['out_idx']] = False
    if not np.all(np.isfinite(Zp)):
        fails.append('predict_Z not finite')
    else:
        r_ = (Zp - Z)[keep]
        r_ = {'both': np.abs(r_), 'real': np.abs(r_.real), 'imag': np.abs(r_.imag)}[kw.get('part', 'both')]   # the fitted part
        rms = float(np.sqrt(np.mean(r_ ** 2)))
        info += ' rms=%.2e' % rms
        # (ten points per spectrum with ten basis functions cannot follow two arcs closely; outliers that the error model was
        #  told to ignore pull the fit)
        slack = (0.03 if len(f) >= 41 else (0.05 if len(f) >= 21 else 0.15)) + (0.05 if case['out_idx'] and kw['outliers'] is False else 0.0)
        if case['out_idx'] and kw['outliers'] is not False:
            slack += 0.01                                     # (the outlier model shares the misfit between sigma_out and the curve)
        if case['mode'] == 'sample':
            slack = max(slack, 0.4)                           # (60 warm-up iterations: a sanity band, not a convergence claim)
        if not rms <= 5 * np.sqrt(2) * case['sig'] + slack * np.mean(np.abs(Z)):
            fails.append('residual rms %.3g (noise %.3g, mean|Z| %.3g)' % (rms, case['sig'], np.mean(np.abs(Z))))
    tau = np.logspace(np.log10(1 / (2 * np.pi * f[0])) - 0.5, np.log10(1 / (2 * np.pi * f[-1])) + 0.5, 120)
    g = inv.predict_distribution('DRT', eval_tau=tau)
    if case['multi']:
        g2 = inv.predict_distribution('TP-DDT', eval_tau=tau)
        if not (np.all(np.isfinite(g2)) and np.min(g2) >= -1e-9 * np.max(np.abs(g2))):
            fails.append('DDT distribution not finite / negative')
    if not np.all(np.isfinite(g)):
        fails.append('gamma not finite')
    elif kw['nonneg'] and np.min(g) < -1e-9 * np.max(np.abs(g)):
        fails.append('negative gamma %.3g in a non-negative fit' % np.min(g))
    tol = (0.25 if len(f) >= 21 and case['mode'] == 'optimize' else 0.6) + 10 * (case['sig'] / np.mean(np.abs(Z)))
    if not (np.isfinite(inv.R_inf) and abs(inv.R_inf - case['R0']) <= tol * (case['R0'] + 0.3 * case['Rp'])):
        fails.append('R_inf %.4g (true %.4g)' % (inv.R_inf, case['R0']))
    if case['multi']:
        fails += _api_sweep(inv, case, tau)
        return ('FAIL', text + ' :: ' + info + '\n    ' + '\n    '.join(fails)) if fails else ('ok', text + ' :: ' + info)
    rp = inv.predict_Rp()
    rp = float(np.atleast_1d(rp)[0]) if np.ndim(rp) else float(rp)
    if not (np.isfinite(rp) and abs(rp - case['Rp']) <= tol * case['Rp'] + 0.3 * case['R0'] * tol):
        fails.append('R_p %.4g (true %.4g)' % (rp, case['Rp']))
    if case['mode'] == 'sample':
        lo, med, hi_ = (inv.predict_distribution('DRT', eval_tau=tau, percentile=q) for q in (2.5, 50, 97.5))
        if not (np.all(lo <= med + 1e-12) and np.all(med <= hi_ + 1e-12) and np.all(np.isfinite(hi_))):
            fails.append('percentile band does not bracket its median')
    fails += _api_sweep(inv, case, tau)
    return ('FAIL', text + ' :: ' + info + '\n    ' + '\n    '.join(fails)) if fails else ('ok', text + ' :: ' + info)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--first', type=int, default=0)
    ap.add_argument('--count', type=int, default=100)
    ap.add_argument('--seconds', type=float, default=0.0)
    a = ap.parse_args()
    t0 = time.time()
    tally = dict(ok=0, FAIL=0)
    for n in range(a.first, a.first + a.count):
        if a.seconds and time.time() - t0 > a.seconds:
            break
        try:
            st, text = run_case(n)
        except Exception as e:
            import traceback
            st, text = 'FAIL', make_case(n)[1] + ' :: exception %s: %s\n%s' % (type(e).__name__, str(e)[:200], traceback.format_exc()[-600:])
        tally[st] += 1
        print('case %4d %-4s %s' % (n, st, text), flush=True)
    print('TOTAL %d ok, %d FAILED in %.0f s' % (tally['ok'], tally['FAIL'], time.time() - t0))
    return 1 if tally['FAIL'] else 0


if __name__ == '__main__':
    sys.exit(main())
