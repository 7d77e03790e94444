"""The other model families of the reference's published HMC study through the drop-in API (tests/golden/hmc_suite2.npz:
Run fits.ipynb cells 10, 14, 18, 22): RC-ZARC (Series_pos, own 51-point basis), trunc (Series, sign-free), BimodalTP-DDT /
BimodalBP-DDT (one parallel diffusion distribution: the `Parallel` model), DRT-k-TpDDT (Series-Parallel_pos),
DRT-TpDDT-BpDDT (Series-2Parallel_pos).  Settings = the notebook's; 2 chains x (200 + 200), seed 1234, random init.
Per spectrum and distribution: posterior mean / 2.5 % / 97.5 % curves against the stored ones, saturated / divergent iterations
(ours | reference).  Usage: hmc_suite2_run.py [stem-substring ...]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load, rel_l2
from bayes_drt_amd.inversion import Inverter

S = load('hmc_suite2')
args = [a for a in sys.argv[1:] if not a.startswith('--') and not a.isdigit()]
opt = lambda name, default: int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
NSEEDS, CHAINS, WARM, DRAWS = opt('--seeds', 1), opt('--chains', 2), opt('--warmup', 200), opt('--draws', 200)
XS2P = float(sys.argv[sys.argv.index('--s2p-xscale') + 1]) if '--s2p-xscale' in sys.argv else None


def setup(stem):
    """(Inverter, fit kwargs, tau_plot, [(distribution name, column prefix)]) for one spectrum, as the notebook builds them."""
    fam = stem.split('_')[0]
    tau_plot = np.logspace(-7, 2, 200)
    sm = 0.005 if 'noiseless' in stem else 0.002
    if fam == 'RC-ZARC':                                   # cells 8, 10
        inv = Inverter(basis_freq=1 / (2 * np.pi * np.logspace(-2, 3, 51)))
        return inv, dict(nonneg=True, sigma_min=0.002), np.logspace(np.log10(np.exp(-5)), np.log10(np.exp(5.5)), 200), [('DRT', 'gamma')]
    if fam == 'trunc':                                     # cells 12, 14
        return Inverter(basis_freq=np.logspace(6, -2, 81)), dict(nonneg=False, sigma_min=sm), tau_plot, [('DRT', 'gamma')]
    if fam in ('BimodalTP-DDT', 'BimodalBP-DDT'):          # cells 16, 18
        bc = 'transmissive' if 'TP' in fam else 'blocking'
        inv = Inverter(distributions={'DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': bc, 'dist_type': 'parallel',
                                              'basis_freq': np.logspace(6, -3, 91)}})
        return inv, dict(sigma_min=sm), tau_plot, [('DDT', 'gamma')]
    if fam.startswith('DRT-') and fam.endswith('-TpDDT'):  # cell 20: sp_dr
        inv = Inverter(distributions={'DRT': {'kernel': 'DRT', 'basis_freq': np.logspace(6, -2, 81)},
                                      'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel',
                                                 'basis_freq': np.logspace(6, -2, 81), 'x_scale': 0.8}})
        return inv, dict(nonneg=True, sigma_min=sm), tau_plot, [('DRT', 'gamma'), ('TP-DDT', 'ftp')]
    xs = {} if XS2P is None else {'x_scale': XS2P}        # (the notebook's comment: "bayes: xp1_scale = 0.5, xp2_scale = 0.5 (1 and 0.8 also work fine)")
    inv = Inverter(distributions={'DRT': {'kernel': 'DRT'},   # cells 20, 22: s2p_dr, basis logspace(6, -2, 81)
                                  'TP-DDT': dict({'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel'}, **xs),
                                  'BP-DDT': dict({'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'blocking', 'dist_type': 'parallel'}, **xs)},
                   basis_freq=np.logspace(6, -2, 81))
    return inv, dict(nonneg=True, sigma_min=sm), tau_plot, [('DRT', 'gamma'), ('TP-DDT', 'ftp'), ('BP-DDT', 'fbp')]


print('%-30s %-32s %5s | %-18s | %-11s | %s' % ('spectrum', 'model', 'secs', 'saturated ours|ref', 'div o|r', 'per distribution: mean / lo / hi rel-L2 vs stored'))
for stem in S['stems']:
    stem = str(stem)
    if args and not any(a in stem for a in args):
        continue
    key = stem.replace('-', '').replace('.', 'p')
    Zd, G, cols, d = S['Z__' + key], S['G__' + key], [str(c) for c in S['Gcols__' + key]], S['diag__' + key]
    f, Z = Zd[:, 0], Zd[:, 1] + 1j * Zd[:, 2]
    means = []
    for sd in range(NSEEDS):
        inv, kw, tau_plot, dists = setup(stem)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            t0 = time.time()
            inv.fit(f, Z, mode='sample', warmup=WARM, samples=DRAWS, chains=CHAINS, random_seed=1234 + 1000 * sd, **kw)
            dt = time.time() - t0
        fit = inv._sample_result
        parts = []
        for name, col in dists:
            g = inv.predict_distribution(name, eval_tau=tau_plot)
            lo = inv.predict_distribution(name, eval_tau=tau_plot, percentile=2.5)
            hi = inv.predict_distribution(name, eval_tau=tau_plot, percentile=97.5)
            parts.append('%s %.4f %.4f %.4f' % (name, rel_l2(g, G[:, cols.index(col)]), rel_l2(lo, G[:, cols.index(col + '_lo')]),
                                                rel_l2(hi, G[:, cols.index(col + '_hi')])))
            if name == dists[0][0]:
                means.append(g)
        print('%-30s %-32s %5.1f | %4d | %-11s | %3d | %-5s | %s ; leapfrogs %d' % (
            stem, inv.stan_model_name.replace('_StanModel.pkl', ''), dt, fit.n_max_treedepth, 'n/a' if np.isnan(d[0]) else int(d[0]),
            fit.n_divergent, 'n/a' if np.isnan(d[1]) else int(d[1]), ' ; '.join(parts), fit.n_leapfrog), flush=True)
    if NSEEDS > 1:
        m = np.array(means)
        print('    our own seed-to-seed scatter of the posterior mean (%d seeds, %d chains x (%d + %d)): rel-L2 between runs %s'
              % (NSEEDS, CHAINS, WARM, DRAWS, ' '.join('%.4f' % rel_l2(m[a], m[b]) for a in range(NSEEDS) for b in range(a + 1, NSEEDS))))
