"""A/B of the MAP iteration's two coordinate choices on the random end-to-end cases of tests/fuzz_inverter.py (single-distribution,
mode='optimize' cases only): coefficients on the linear scale from lam <= 1e-4 on (round 6, default) against the log scale throughout
(BDRT_NEWTON_LINEAR=0, the iteration of rounds 1-5), one start each (n_starts=1) so that what is compared is the iteration itself.
Per case: log-posterior reached, rounds, |grad|_inf, return code.  Usage: map_scale_ab.py [first] [count]   (runs itself twice as child processes)"""
import json, os, subprocess, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(first, count):
    from tests.fuzz_inverter import make_case
    from bayes_drt_amd.inversion import Inverter
    out = {}
    for n in range(first, first + count):
        case, text = make_case(n)
        if case['mode'] != 'optimize' or case['multi']:
            continue
        kw = dict(case['kw'])
        if kw.get('outliers'):
            continue                                       # (the outlier models keep the finite-difference Hessian on either side)
        kw['outliers'] = False
        inv = Inverter(basis_freq=case['bf'])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            try:
                inv.fit(case['f'], case['Z'], mode='optimize', n_starts=1, **kw)
                r = inv._opt_report
                out[n] = dict(lp=r['lp'], it=r['newton_iterations'], ginf=r['grad_inf'], rc=r['return_code'], evals=r['n_evals'], text=text[:60])
            except Exception as e:
                out[n] = dict(error=str(e)[:80])
    print('RESULT ' + json.dumps(out))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        run(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    res = {}
    for tag, env in (('linear', {}), ('log', {'BDRT_NEWTON_LINEAR': '0'})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', str(first), str(count)], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')]
        res[tag] = json.loads(line[0][7:]) if line else {}
        if not line:
            print(p.stderr[-2000:])
    a, b = res['linear'], res['log']
    keys = sorted(set(a) & set(b), key=int)
    print('# cases %d (single distribution, mode=optimize, no outlier model, one start); columns: case | lp linear / log | rounds linear / log | rc' % len(keys))
    better = worse = same = 0
    it_a = it_b = 0
    for k in keys:
        x, y = a[k], b[k]
        if 'error' in x or 'error' in y:
            print(k, 'ERROR', x.get('error'), y.get('error')); continue
        d = x['lp'] - y['lp']
        tol = 1e-6 * max(1.0, abs(y['lp']))
        flag = 'same' if abs(d) <= tol else ('LINEAR HIGHER by %.3g' % d if d > 0 else 'LOG HIGHER by %.3g' % -d)
        same += abs(d) <= tol; better += d > tol; worse += d < -tol
        it_a += x['it']; it_b += y['it']
        if flag != 'same' or x['rc'] != y['rc']:
            print('%s | %.6f / %.6f | %d / %d | rc %d / %d | %s | %s' % (k, x['lp'], y['lp'], x['it'], y['it'], x['rc'], y['rc'], flag, x['text']))
    print('# same stationary point (lp to 1e-6 relative): %d; linear scale ends higher: %d; log scale ends higher: %d; Newton rounds in all: linear %d, log %d' % (same, better, worse, it_a, it_b))
