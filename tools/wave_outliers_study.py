"""Statistical check of the one-chain-per-wave kernel with the outlier error model in the shape it is used: a batch of spectra x 4 chains
through Inverter.fit_many(outliers=True) at mid occupancy (100 noise realisations of the 2-ZARC spectrum x 4 chains = 400 units: the wave
kernel by default), against the same call with BDRT_WAVE=0 (the general one-chain-per-workgroup kernel takes this many units then, in two turns).  Long chains are not expected to agree
draw by draw (another summation order); the posterior means of the two runs agree within their Monte-Carlo error, which the difference
between the two halves of the chains of ONE run measures.
Usage: wave_outliers_study.py [n_spectra]"""
import os, sys, time, warnings, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from tests.helpers import load, rel_l2
    from bayes_drt_amd.inversion import Inverter
    n = int(sys.argv[2])
    c = load('csv_2ZARC_uniform_0.25')
    f, Z0 = c['Z'][:, 0], c['Z'][:, 1] + 1j * c['Z'][:, 2]
    rng = np.random.default_rng(5)
    Zs = []
    for i in range(n):
        Z = Z0 * (1 + 0.002 * rng.standard_normal(len(f))) + 0.002 * np.abs(Z0).mean() * (rng.standard_normal(len(f)) + 1j * rng.standard_normal(len(f)))
        for k in rng.choice(len(f), 2, replace=False):
            Z[k] *= 1.3                                      # two outliers per spectrum
        Zs.append(Z)
    tau = np.logspace(-7, 2, 120)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.time()
        views = Inverter(basis_freq=f).fit_many(f, Zs, nonneg=True, outliers=True, mode='sample', warmup=150, samples=150, chains=4, random_seed=int(os.environ.get('STUDY_SEED', 11)))
        dt = time.time() - t0
    out = {'wall': dt, 'mean': [], 'half': [], 'leap': 0, 'div': 0}
    for v in views:
        fit = v._sample_result
        x = fit.chain_draws('x')                               # [chains, draws, K]
        out['mean'].append(x.mean(axis=(0, 1)).tolist())
        out['half'].append(float(rel_l2(x[:2].mean(axis=(0, 1)), x[2:].mean(axis=(0, 1)))))
        out['leap'] += fit.n_leapfrog; out['div'] += fit.n_divergent
    print(json.dumps(out))
    sys.exit(0)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
res = {}
for tag, env in (('default (one chain per wave)', {}), ('BDRT_WAVE=0 (general one-chain kernel)', {'BDRT_WAVE': '0'})):
    p = subprocess.run([sys.executable, __file__, 'child', str(n)], env=dict(os.environ, **env), capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert line, p.stderr[-2000:]
    res[tag] = json.loads(line[-1])
    r = res[tag]
    print('%-32s %d spectra x 4 chains x (150 + 150): %.1f s wall, %.1f M leapfrogs, %d divergent; median difference between the two halves of a '
          'run\'s chains (posterior mean of x, rel-L2): %.4f' % (tag, n, r['wall'], r['leap'] / 1e6, r['div'], float(np.median(r['half']))))
a, b = (np.array(res[k]['mean']) for k in res)
d = np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
print('posterior mean of x, one run against the other, per spectrum: median rel-L2 %.4f, 90th percentile %.4f, max %.4f' % (np.median(d), np.percentile(d, 90), d.max()))
