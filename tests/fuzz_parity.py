"""Randomised parity cases: problems of random shape (frequencies 6 ... 128, basis 12 ... 192, one to three distributions,
series / parallel, sign constraint, both outlier error models, log-uniform and irregular frequency lists, one or several
spectra) are evaluated and sampled on the GPU and checked against the oracle.

Test infrastructure (it imports oracle/).  `tests/test_gpu_fuzz.py` runs a fixed list of case numbers under `-m gpu`;
`python -m tests.fuzz_parity --first 0 --count 300` is the soak that list was drawn from (record: profiles/r02/fuzz_parity.txt).

Per case:
  0. the matrices built on the device (bdrt_build_A / _L / _M) against the oracle's: the case's own A and L (1e-11 of the
     largest entry), plus one A of a random kernel (DRT, blocking / transmissive planar DDT, series / parallel, charge transfer) and M_0..2;
  1. log-posterior and gradient at random points, with and without the Jacobian term: |dlp| <= 1e-10 max(1, |lp|),
     |dg|_inf <= 1e-10 max(1, |g|_inf);
  2. the one-chain-per-workgroup evaluator of bdrt_solo_wide.h (where the problem takes it): same bounds;
  3. a short NUTS run (0 / 6 / 20 / 24 warm-up iterations + 4 draws, tree depth <= 5) of 1 ... 40 units: per checked unit the number of leapfrogs and of
     divergences equal the oracle's, the draws agree to 1e-6 of the largest coordinate (1e-4 after 20 / 24 warm-up iterations:
     summation-order noise amplified by the leapfrogs; bit-equality is not defined between a tree-reduced and a sequential sum).
     A 24-28-transition run of a chain with divergent transitions may leave the oracle's trajectory altogether (1-2 % of the
     cases, never the short runs): the case is then re-run with 6 warm-up iterations, must agree there, and is marked;
  4. MAP from a random start (bdrt_optimize, defaults): where convergence is reported (|grad|_inf < 1e-8), the oracle's gradient
     at the answer is < 1e-6 and its lp equals the reported lp to 1e-9; the lp never ends below the start's.
"""
import argparse
import ctypes as C
import sys
import time

import numpy as np


def make_case(n):
    """Case number -> (constructor arguments, description).  Pure function of n."""
    from oracle import oracle as orc
    rng = np.random.default_rng(1000003 * n + 17)
    nf = int(rng.choice([6, 9, 16, 23, 31, 40, 57, 64, 71, 81, 96, 107, 128, int(rng.integers(6, 129))]))
    decades = float(rng.uniform(3.0, 9.0))
    f_hi = 10 ** float(rng.uniform(3.0, 7.0))
    f = np.logspace(np.log10(f_hi), np.log10(f_hi) - decades, nf)
    irregular = rng.random() < 0.2
    if irregular:
        f = np.sort(f * np.exp(0.3 * decades / nf * rng.standard_normal(nf)))[::-1].copy()
    nblocks = int(rng.choice([1, 1, 1, 2, 2, 3]))
    own_spacing = rng.random() < 0.25                    # basis spacing unrelated to the frequency spacing: A is not Toeplitz
    # every tenth case from 1000 on: a shape of the Toeplitz-table evaluator (one block, nf = 80..82, K = 80..82 or 160..162 on the frequencies' own
    # log-uniform grid); drawn from a generator of its own so that the other cases keep their numbers
    ta_K = 0
    if n >= 1000 and n % 10 == 7:
        r2 = np.random.default_rng(7000001 * n + 3)
        nf = int(r2.choice([80, 81, 82])); ta_K = int(r2.choice([80, 81, 82, 160, 161, 162]))
        f = np.logspace(np.log10(f_hi), np.log10(f_hi) - decades, nf)
        irregular, own_spacing, nblocks = False, False, 1
    blocks = []
    desc = []
    mats = []
    for b in range(nblocks):
        ext_lo, ext_hi = float(rng.uniform(0.0, 2.0)), float(rng.uniform(0.0, 2.0))
        if own_spacing or irregular:
            K = int(rng.integers(12, 193 if nblocks == 1 else 130))
            bf = np.logspace(np.log10(f[0]) + ext_hi, np.log10(f[-1]) - ext_lo, K)
        else:
            step = decades / (nf - 1)
            n_hi, n_lo = int(round(ext_hi / step)), int(round(ext_lo / step))
            if ta_K:
                n_hi = max(ta_K - nf, 0) // 2; n_lo = max(ta_K - nf, 0) - n_hi
            K = nf + n_hi + n_lo
            cap = 192 if nblocks == 1 else 129
            while K > cap:
                if n_hi >= n_lo and n_hi > 0: n_hi -= 1
                elif n_lo > 0: n_lo -= 1
                else: break
                K = nf + n_hi + n_lo
            if K > cap:
                K = cap
                bf = np.logspace(np.log10(f[0]), np.log10(f[-1]), K)
            else:
                bf = 10 ** (np.log10(f[0]) + step * n_hi - step * np.arange(K))
        tau = 1 / (2 * np.pi * bf)
        eps = 1 / np.mean(np.diff(np.log(tau))) * float(rng.choice([1.0, 1.0, 0.7]))
        parallel = b > 0
        if parallel:
            akw = dict(kernel='DDT', dist_type='parallel', symmetry='planar', bc=str(rng.choice(['transmissive', 'blocking'])))
        else:
            akw = dict(kernel='DRT', dist_type='series')
        A = np.vstack([orc.construct_A(f, 'real', tau=tau, epsilon=eps, **akw),
                       orc.construct_A(f, 'imag', tau=tau, epsilon=eps, **akw)])
        L = [orc.construct_L(tau, eps, o) for o in (0, 1, 2)]
        mats.append((bf, tau, eps, akw, A, L))
        nonneg = True if parallel else bool(rng.random() < 0.7)
        blocks.append(dict(A=A, L0=float(rng.choice([1.0, 0.5])) * L[0], L1=L[1], L2=float(rng.choice([0.75, 0.5, 1.0])) * L[2],
                           parallel=parallel, nonneg=nonneg, x_scale=float(np.exp(rng.uniform(-1, 1))) if parallel else 1.0))
        desc.append('%s%s K=%d' % ('P' if parallel else 'S', '+' if nonneg else '', K))
    # a spectrum: a two-peak series distribution + offset + noise (a physical fit is not needed for parity)
    A0 = blocks[0]['A']
    K0 = A0.shape[1]
    lt = np.linspace(-1, 1, K0)
    x = np.exp(-0.5 * ((lt - rng.uniform(-0.6, 0.6)) / 0.15) ** 2) + 0.5 * np.exp(-0.5 * ((lt - rng.uniform(-0.6, 0.6)) / 0.3) ** 2)
    n_spectra = int(rng.choice([1, 1, 1, 3]))
    Zs = []
    for s in range(n_spectra):
        Z = A0 @ (x * (1 + 0.2 * s)) + np.concatenate([np.full(nf, float(rng.uniform(0.1, 1.0))), np.zeros(nf)])
        Z = Z / np.std(np.hypot(Z[:nf], Z[nf:])) + 0.01 * rng.standard_normal(2 * nf)
        if rng.random() < 0.3:
            Z[int(rng.integers(0, 2 * nf))] += 0.3                # an outlier
        Zs.append(Z)
    om = int(rng.choice([0, 0, 0, 1])) if nblocks == 1 else int(rng.choice([0, 0, 2]))
    kw = dict(sigma_min=float(rng.choice([0.002, 0.01])), ups_alpha=float(rng.choice([1.0, 0.05])), ups_beta=float(rng.choice([0.1, 0.8])),
              induc_scale=float(rng.choice([1.0, 0.3])), outlier_mode=om)
    if om == 1:
        kw.update(so_lambda=float(rng.choice([10.0, 5.0])), so_alpha=float(rng.choice([5.0, 2.0])), so_beta=1.0)
    elif om == 2:
        kw.update(so_lambda=float(rng.choice([10.0, 4.0])))
    if nblocks > 1:
        kw.update(use_x_sum=True, x_sum_invscale=float(rng.choice([0.0, 0.5])))
    n_units = int(rng.choice([1, 2, 4, 5, 16, 17, 33, 40]))
    text = 'nf=%d%s %s outl=%d spectra=%d units=%d' % (nf, ' irregular' if irregular else (' own-spacing' if own_spacing else ''),
                                                       ' | '.join(desc), om, n_spectra, n_units)
    return dict(blocks=blocks, Z=np.array(Zs), freq=f, kw=kw, n_units=n_units, seed=int(rng.integers(1, 10 ** 6)), mats=mats), text


def _wide1(prob, theta, jac):
    fn = prob._lib.bdrt_debug_wide1_logp_grad
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lp = np.empty(len(theta)); g = np.empty_like(theta)
    rc = fn(prob.handle, theta.ctypes.data, None, len(theta), int(jac), lp.ctypes.data, g.ctypes.data)
    if rc == -2:
        return None, None
    assert rc == 0, prob._lib.bdrt_last_error().decode()
    return lp, g


def run_case(n, verbose=False):
    """Returns (status, text): status 'ok', 'skip' (the problem is beyond the LDS budget or the sampler's D <= 864: loud error,
    by design) or 'FAIL'."""
    from bayes_drt_amd._lib import BdrtError, NutsControl
    from bayes_drt_amd.engine import Sampler
    from bayes_drt_amd.model import Problem
    from oracle import oracle as orc
    case, text = make_case(n)
    try:
        prob = Problem(case['blocks'], case['Z'], case['freq'], **case['kw'])
    except BdrtError as e:
        if 'LDS' in str(e):
            return 'skip', text + ' :: ' + str(e)[:60]
        raise
    evaluator = prob.evaluator()
    n_spectra = len(case['Z'])
    oms = [orc.OracleModel(case['blocks'], case['Z'][s], case['freq'], **case['kw']) for s in range(n_spectra)]
    rng = np.random.default_rng(n)
    fails = []
    # 0: the device-built matrices (bdrt_build_A / _L) against the oracle's, 1e-11 of the largest entry
    from bayes_drt_amd import matrices as gm
    for b, (bf, tau, eps, akw, A, L) in enumerate(case['mats']):
        Ag = np.vstack([gm.construct_A(case['freq'], p, tau=tau, epsilon=eps, **akw) for p in ('real', 'imag')])
        if not np.max(np.abs(Ag - A)) <= 1e-11 * np.max(np.abs(A)):
            fails.append('block %d: A differs by %.3g (scale %.3g)' % (b, np.max(np.abs(Ag - A)), np.max(np.abs(A))))
        for o in (0, 1, 2):
            Lg = gm.construct_L(bf, tau=tau, epsilon=eps, order=o)
            if not np.max(np.abs(Lg - L[o])) <= 1e-11 * np.max(np.abs(L[o])):
                fails.append('block %d: L%d differs by %.3g' % (b, o, np.max(np.abs(Lg - L[o]))))
    # 0b: one more matrix set per case with a random kernel (all DDT kinds, series / parallel, charge transfer) and the
    #     integrated-penalty matrices M_0..2, on the case's own frequency / tau grids
    rk = np.random.default_rng(7 * n + 3)
    # (not the blocking-spherical DDT: the reference's formula tanh x / (x - tanh x) loses every digit for omega tau << 1 --
    #  on these wide random grids the reference-style evaluation gives inf / NaN / noise in both implementations, each in its
    #  own way; tests/test_gpu_matrices.py checks that kernel against the reference's own matrices where those are finite)
    kind = [('DRT', None, None), ('DDT', 'blocking', 'planar'), ('DDT', 'transmissive', 'planar'), ('DDT', 'transmissive', 'planar')][int(rk.integers(0, 4))]
    bf0, tau0, eps0 = case['mats'][0][0], case['mats'][0][1], case['mats'][0][2]
    akw = dict(kernel=kind[0], dist_type='series' if kind[0] == 'DRT' else str(rk.choice(['series', 'parallel'])))
    if kind[0] == 'DDT':
        akw.update(bc=kind[1], symmetry=kind[2])
        if rk.random() < 0.4:
            akw.update(ct=True, k_ct=float(10 ** rk.uniform(-2, 3)))
    try:
        if len(case['freq']) * len(tau0) > 5000 and n % 4:
            raise RuntimeError('skipped')                  # (the CPU quadrature of a large general matrix takes seconds)
        Ao = [orc.construct_A(case['freq'], p, tau=tau0, epsilon=eps0, **akw) for p in ('real', 'imag')]
        Ag = [gm.construct_A(case['freq'], p, tau=tau0, epsilon=eps0, **akw) for p in ('real', 'imag')]
        mag = np.sqrt(Ao[0] ** 2 + Ao[1] ** 2)
        okm = np.isfinite(Ao[0]) & np.isfinite(Ao[1])
        err = np.maximum(np.abs(Ag[0] - Ao[0]), np.abs(Ag[1] - Ao[1]))
        tolA = 1e-11
        if not (np.all(np.isfinite(Ag[0][okm])) and np.all(err[okm] <= tolA * mag[okm] + 1e-300)):
            fails.append('matrix %s: relative error %.3g' % (akw, float(np.nanmax(err[okm] / mag[okm]))))
    except Exception as e:
        if 'not equal' not in str(e) and str(e) != 'skipped':   # (the reference's own Toeplitz consistency check may refuse a grid)
            fails.append('matrix %s: %s %s' % (akw, type(e).__name__, e))
    for o in (0, 1, 2):
        Mo, Mg = orc.construct_M(tau0, eps0, o), gm.construct_M(bf0, order=o, epsilon=eps0)
        if not np.max(np.abs(Mg - Mo)) <= 1e-11 * np.max(np.abs(Mo)):
            fails.append('M%d differs by %.3g of the largest entry' % (o, np.max(np.abs(Mg - Mo)) / np.max(np.abs(Mo))))
    # 1, 2: evaluators
    npts = 5
    theta = np.ascontiguousarray(rng.uniform(-2, 2, (npts, prob.D)))
    spec = (np.arange(npts) % n_spectra).astype(np.int32)
    for jac in (True, False):
        lp, g = prob.logp_grad(theta, jacobian=jac, spec=spec)
        w_lp, w_g = (None, None) if n_spectra > 1 else _wide1(prob, theta, jac)
        for i in range(npts):
            lr, gr = oms[spec[i]].logp_grad(theta[i], jac)
            for nm, a, b in (('batched', lp, g), ('one-chain', w_lp, w_g)):
                if a is None:
                    continue
                if not np.isfinite(lr):
                    if not (a[i] == lr or (np.isnan(a[i]) and np.isnan(lr))):
                        fails.append('%s lp non-finite mismatch %r vs %r' % (nm, a[i], lr))
                    continue
                if not abs(a[i] - lr) <= 1e-10 * max(1.0, abs(lr)):
                    fails.append('%s lp jac=%d pt %d: %.17g vs %.17g' % (nm, jac, i, a[i], lr))
                e = np.max(np.abs(b[i] - gr))
                if not e <= 1e-10 * max(1.0, np.max(np.abs(gr))):
                    fails.append('%s grad jac=%d pt %d: err %.3g at %d (|g| %.3g)' % (nm, jac, i, e, int(np.argmax(np.abs(b[i] - gr))),
                                                                                     np.max(np.abs(gr))))
    # 3: sampler
    ctrl = NutsControl(); prob._lib.bdrt_nuts_defaults(C.byref(ctrl)); ctrl.max_treedepth = 5
    # (warm-up lengths on both sides of Stan's 20-iteration threshold for windowed adaptation: 20 and 24 include a metric
    #  update and the step-size search that follows it)
    warm = int(np.random.default_rng(11 * n + 1).choice([6, 6, 0, 20, 24]))
    nd, nu, seed = 4, case['n_units'], case['seed']
    uspec = (np.arange(nu) % n_spectra).astype(np.int32)
    def run_and_compare(warm_):
        smp = Sampler(prob, nu, warm_, nd, seed, ctrl, spec=uspec)
        with smp:
            kind_ = smp.kind()
            smp.run(None)
            draws, lps, diag = smp.results()
        bad = []
        for c in sorted({0, nu // 2, nu - 1}):
            ref, lpr, dr = orc.nuts_sample(oms[uspec[c]], c, seed, warm_, nd, control=orc.nuts_control(max_treedepth=5))
            if dr['n_leapfrog'] != diag[c]['n_leapfrog'] or dr['n_divergent'] != diag[c]['n_divergent']:
                bad.append('sampler unit %d: leapfrogs %d vs %d, divergences %d vs %d' % (c, diag[c]['n_leapfrog'], dr['n_leapfrog'],
                                                                                          diag[c]['n_divergent'], dr['n_divergent']))
                continue
            err = np.max(np.abs(draws[c] - ref)) / np.max(np.abs(ref))
            if not err < (1e-6 if warm_ <= 6 else 1e-4):      # (28 transitions amplify the summation-order noise further)
                bad.append('sampler unit %d: draws differ by %.3g' % (c, err))
        return kind_, bad
    try:
        kind, bad = run_and_compare(warm)
    except BdrtError as e:
        if 'not supported' in str(e) and prob.D > 864:      # three distributions + the outlier model: no reference family
            prob.close()
            return 'skip', text + ' D=%d :: %s' % (prob.D, str(e)[-40:])
        raise
    note = ''
    if bad and warm > 6:
        # A chain on a rough posterior (divergent transitions, step size not yet adapted) amplifies the summation-order noise
        # by ~e^(0.06 leapfrogs): after 24-28 transitions a tree decision can flip, and everything after it differs.  That is
        # the oracle's own sensitivity, not a property of the kernels -- provided the same units agree over the short run.
        kind, bad6 = run_and_compare(6)
        if bad6:
            fails += bad + ['(and at 6 warm-up iterations) ' + x for x in bad6]
        else:
            note = ' [long run left the oracle\'s trajectory: ' + bad[0] + '; the 6-iteration run of the same units agrees]'
    else:
        fails += bad
    # 4: MAP (Levenberg-Marquardt Newton on the device): where it reports convergence the oracle's gradient at the answer
    #    vanishes as well and the oracle's lp equals the reported one; "no convergence" is reported, never hidden
    from bayes_drt_amd.engine import optimize_batch
    th0 = np.random.default_rng(n + 7).uniform(-2, 2, (1, prob.D))
    out, rep = optimize_batch(prob, th0, max_iter=2000)
    lr, gr = oms[0].logp_grad(out[0], False)
    conv = rep[0]['return_code'] == 0 and rep[0]['grad_inf'] < 1e-8
    if conv and not (np.max(np.abs(gr)) < 1e-6 and abs(lr - rep[0]['lp']) <= 1e-9 * max(1.0, abs(lr))):
        fails.append('MAP: reported |g| %.3g lp %.12g, oracle |g| %.3g lp %.12g' % (rep[0]['grad_inf'], rep[0]['lp'], np.max(np.abs(gr)), lr))
    boundary = False
    if not np.isfinite(lr) and case['kw'].get('use_x_sum') and np.isfinite(rep[0]['lp']):
        # The mixed models reject x_sum_raw < 0 (real<lower=0> x_sum_raw): with a series block without sign constraint the optimum can lie
        # ON that boundary.  There the device's iterate is feasible by its own summation order and infeasible by the oracle's (the sum is
        # zero to rounding), the iteration stalls against the wall and says so (return code 2): reported, not a parity failure.
        lay = prob.layout()
        terms = np.concatenate([np.exp(out[0][o:o + blk['A'].shape[1]]) if blk.get('nonneg') else out[0][o:o + blk['A'].shape[1]]
                                for o, blk in zip(lay['x'], case['blocks'])])
        boundary = abs(terms.sum()) <= 1e-9 * np.abs(terms).sum() and rep[0]['return_code'] != 0
    if boundary:
        note += ' [MAP stalled on the boundary x_sum_raw = 0 of the support (return code %d)]' % rep[0]['return_code']
    elif not np.isfinite(lr) or lr < oms[0].logp_grad(th0[0], False)[0]:
        fails.append('MAP: lp at the answer %.6g below lp at the start' % lr)
    prob.close()
    text += note + ' D=%d evaluator=%d kernel=%d warm=%d map=%s' % (prob.D, evaluator, kind, warm, 'converged/%d' % rep[0]['newton_iterations'] if conv else
                                        'rc%d,|g|=%.1e' % (rep[0]['return_code'], rep[0]['grad_inf']))
    if fails:
        return 'FAIL', text + '\n    ' + '\n    '.join(fails)
    return 'ok', text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--first', type=int, default=0)
    ap.add_argument('--count', type=int, default=100)
    ap.add_argument('--seconds', type=float, default=0.0, help='stop after this much wall time (0: run all)')
    a = ap.parse_args()
    t0 = time.time()
    tally = dict(ok=0, skip=0, FAIL=0)
    kinds = {}
    for n in range(a.first, a.first + a.count):
        if a.seconds and time.time() - t0 > a.seconds:
            break
        try:
            st, text = run_case(n)
        except Exception as e:                              # a crash is a failure of the case, keep going
            st, text = 'FAIL', 'exception %s: %s' % (type(e).__name__, str(e)[:200])
        tally[st] += 1
        print('case %4d %-4s %s' % (n, st, text), flush=True)
    print('TOTAL %d ok, %d skipped (beyond the LDS budget / D > 864: loud errors), %d FAILED in %.0f s' % (tally['ok'], tally['skip'], tally['FAIL'],
                                                                                 time.time() - t0))
    return 1 if tally['FAIL'] else 0


if __name__ == '__main__':
    sys.exit(main())
