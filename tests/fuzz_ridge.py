"""Randomised cases of the ridge row: `Inverter.ridge_fit` on random spectra (10 ... 128 frequencies, default or extended basis,
random penalty / part / sign constraint / weights / lambda_0 / f_beta), the device-resident hyper-lambda loop (bdrt_ridge)
against the same iteration driven from the host around the batched QP kernel, and the optimality of the answer itself.

Test infrastructure.  `python -m tests.fuzz_ridge --count 300`; tests/test_gpu_fuzz.py runs a fixed slice under -m gpu.

Per case:
  1. same number of hyper-lambda iterations, coefficients within 1e-6 of the largest coefficient (1e-4 when the iteration
     stopped at max_iter without converging), R_inf within 1e-6;
  2. the fitted impedance is finite and follows the spectrum (rms residual < 0.5 mean |Z|: a sanity band, the fit quality at
     10 frequencies and lambda_0 ~ 1 is what it is);
  4. every eighth case: `ridge_ReImCV` over 8 lambdas -- the batched launch gives bit for bit the CV table of the sequential
     loop;
  3. ordinary ridge (hyper_lambda=False): the KKT conditions of the QP 1/2 x'Px + q'x, x >= lo at the answer of the batched
     interior-point kernel: multipliers >= -1e-5 max|q|, complementarity gap <= 2e-6 |objective| + 1e-7 (cvxopt's tolerances).
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np


def make_case(n):
    rng = np.random.default_rng(7919 * n + 3)
    nf = int(rng.choice([10, 21, 41, 61, 81, 101, 128, int(rng.integers(10, 129))]))
    hi, lo = float(rng.uniform(4, 7)), float(rng.uniform(-3, 0))
    f = np.logspace(hi, lo, nf)
    w = 2 * np.pi * f
    # two ZARC elements + series resistance + inductance
    R0, L = float(rng.uniform(0.1, 2.0)), float(10 ** rng.uniform(-8, -6))
    Z = R0 + 1j * w * L
    for _ in range(2):
        R, t0, a = float(rng.uniform(0.3, 3.0)), float(10 ** rng.uniform(-hi + 1, -lo - 1)), float(rng.uniform(0.6, 1.0))
        Z = Z + R / (1 + (1j * w * t0) ** a)
    sig = float(rng.choice([0.0, 0.002, 0.01])) * np.mean(np.abs(Z))
    Z = Z + sig * (rng.standard_normal(nf) + 1j * rng.standard_normal(nf))
    ext = bool(rng.random() < 0.4)
    if ext:
        ppd = (nf - 1) / (hi - lo)
        n_ext = int(rng.integers(1, max(2, int(1.5 * ppd))))
        K = nf + 2 * n_ext
        if K > 201:
            n_ext = (201 - nf) // 2
            K = nf + 2 * n_ext
        bf = np.logspace(hi + n_ext / ppd, lo - n_ext / ppd, K)
    else:
        bf = None
    penalty = str(rng.choice(['discrete', 'discrete', 'integral', 'cholesky']))
    kw = dict(penalty=penalty, part=str(rng.choice(['both', 'both', 'real', 'imag'])), nonneg=bool(rng.random() < 0.8),
              weights=rng.choice([None, 'modulus', 'unity']), lambda_0=float(10 ** rng.uniform(-4, 0)),
              reg_ord=int(rng.choice([2, 2, 1])), hyper_lambda=bool(rng.random() < 0.8))
    if penalty == 'integral':
        kw['hl_beta'] = float(rng.choice([2.5, 3.0]))
    elif rng.random() < 0.3:
        kw['hl_fbeta'] = float(rng.choice([0.1, 0.5]))
    if penalty == 'cholesky':
        kw.pop('hl_fbeta', None)
    text = 'nf=%d basis=%s %s' % (nf, 'default' if bf is None else 'K=%d' % len(bf),
                                  ' '.join('%s=%s' % (k, ('%.2e' % v) if isinstance(v, float) else v) for k, v in kw.items()))
    return dict(f=f, Z=Z, bf=bf, kw=kw, sig=sig), text


def run_case(n):
    from bayes_drt_amd.inversion import Inverter
    case, text = make_case(n)
    f, Z, kw = case['f'], case['Z'], case['kw']
    fails, res = [], []
    for host in (False, True):
        if host:
            os.environ['BDRT_HOST_LAMBDA_LOOP'] = '1'
        else:
            os.environ.pop('BDRT_HOST_LAMBDA_LOOP', None)
        try:
            inv = Inverter(basis_freq=case['bf'])
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                inv.ridge_fit(f, Z, **kw)
            coef = inv.distribution_fits['DRT']['coef'].copy()
            n_it = len(inv._iter_history) if kw['hyper_lambda'] else 1
            res.append((coef, float(inv.R_inf), n_it, inv.predict_Z(f)))
        finally:
            os.environ.pop('BDRT_HOST_LAMBDA_LOOP', None)
    (ca, ra, na, za), (cb, rb, nb, zb) = res
    if na != nb:
        fails.append('iterations: device %d, host %d' % (na, nb))
    scale = max(np.max(np.abs(cb)), 1e-300)
    # (an iteration that ran into max_iter has not contracted: QP-tolerance differences grow through 20 lambda updates)
    tol = 1e-6 if na < kw.get('max_iter', 20) else 1e-4
    if not np.max(np.abs(ca - cb)) <= tol * scale:
        fails.append('coefficients differ by %.3g of the largest' % (np.max(np.abs(ca - cb)) / scale))
    if not abs(ra - rb) <= 1e-6 * max(abs(rb), np.mean(np.abs(Z))):
        fails.append('R_inf %.10g vs %.10g' % (ra, rb))
    if not np.all(np.isfinite(za)):
        fails.append('fitted impedance not finite')
    else:
        part = kw['part']
        r = za - Z
        rr = {'both': np.abs(r), 'real': np.abs(r.real), 'imag': np.abs(r.imag)}[part]
        rms = float(np.sqrt(np.mean(rr ** 2)))
        if not rms <= 0.5 * np.mean(np.abs(Z)):
            fails.append('fit rms %.3g (noise %.3g, |Z| %.3g)' % (rms, case['sig'], np.mean(np.abs(Z))))
    if not kw['hyper_lambda'] and kw['penalty'] != 'cholesky':
        inv = Inverter(basis_freq=case['bf'])
        st = inv._ridge_setup(f, Z, kw['part'], kw['penalty'], kw['reg_ord'], 0, True, kw['nonneg'], kw['weights'], False)
        from bayes_drt_amd.inversion import _qp_batch
        P = st['G'] + kw['lambda_0'] * st['base'][kw['reg_ord']]
        if True:
            q = -st['g']
            x, obj = _qp_batch(P[None], q[None], st['lo'])
            g = P @ x[0] + q
            lo = np.broadcast_to(st['lo'], x[0].shape)
            # an interior-point answer: the multipliers g are >= 0 up to the solver's feasibility tolerance and the
            # complementarity gap sum g (x - lo) is within its gap tolerance (cvxopt's defaults: 1e-7 abs, 1e-6 rel)
            if not np.all(g > -1e-5 * np.max(np.abs(q))):
                fails.append('negative multiplier %.3g (|q| %.3g)' % (np.min(g), np.max(np.abs(q))))
            gap = float(np.sum(np.abs(g * (x[0] - lo))))
            if not gap <= 2e-6 * abs(obj[0]) + 1e-7:
                fails.append('complementarity gap %.3g (objective %.6g)' % (gap, obj[0]))
    if n % 8 == 0 and kw['penalty'] != 'cholesky':
        # 4. Re-Im cross-validation: all 2 x len(lambdas) fits as one launch against the reference-style sequential loop
        cvkw = {k: v for k, v in kw.items() if k in ('penalty', 'nonneg', 'weights', 'reg_ord', 'hl_beta', 'hl_fbeta')}
        lams = np.logspace(-6, 1, 8)
        tabs = []
        for seq in (False, True):
            if seq:
                os.environ['BDRT_SEQUENTIAL_CV'] = '1'
            try:
                inv = Inverter(basis_freq=case['bf'])
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    best = inv.ridge_ReImCV(f, Z, lambdas=lams, **cvkw)
                tabs.append((best, inv.cv_result['recv'].copy(), inv.cv_result['imcv'].copy()))
            finally:
                os.environ.pop('BDRT_SEQUENTIAL_CV', None)
        if tabs[0][0] != tabs[1][0] or not (np.array_equal(tabs[0][1], tabs[1][1]) and np.array_equal(tabs[0][2], tabs[1][2])):
            fails.append('Re-Im CV: batched table differs from the sequential loop (best %g vs %g, max diff %.3g)' % (
                tabs[0][0], tabs[1][0], max(np.max(np.abs(tabs[0][1] - tabs[1][1])), np.max(np.abs(tabs[0][2] - tabs[1][2])))))
        text += ' cv=%.0e' % tabs[0][0]
    text += ' iters=%d' % na
    return ('FAIL', text + '\n    ' + '\n    '.join(fails)) if fails else ('ok', text)


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
            st, text = 'FAIL', make_case(n)[1] + ' :: exception %s: %s' % (type(e).__name__, str(e)[:200])
        tally[st] += 1
        print('case %4d %-4s %s' % (n, st, text), flush=True)
    print('TOTAL %d ok, %d FAILED in %.0f s' % (tally['ok'], tally['FAIL'], time.time() - t0))
    return 1 if tally['FAIL'] else 0


if __name__ == '__main__':
    sys.exit(main())
