"""Generate the golden fixtures under tests/golden/ from the reference (run in the build container only).

The reference package (/root/reference) is imported here -- and only here -- to produce input/output
VECTORS (numpy arrays).  No reference source travels: the fixtures are data.  Re-run with

    python tests/golden/make_golden.py

What is produced:
  mat_*.npz      construct_A / construct_L / construct_M outputs           (bayes_drt/matrices.py)
  ddt_*.npz      construct_A for the DDT kernels (Toeplitz path and general path)
  dat_*.npz      Inverter._prep_matrices + _prep_stan_data output ("dat" dict) (bayes_drt/inversion.py:1684, :2127)
  kat_*.npz      the 37 stored Stan `optimizing` results (code_EchemActa/map_results/obj_*.pkl):
                 parameters + transformed parameters + the matrices they were computed from
  csv_*.npz      reference data/result CSV files as arrays (data/simulated, map_results, bayes_results)
  predict_*.npz  predict_distribution / predict_Z / predict_sigma outputs for given coefficients
"""
import glob
import os
import pickle
import sys
import types
import warnings

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings('ignore')


def _install_stubs():
    # pystan / cvxopt are not installed; only the pre-Stan Python of the reference is executed.
    cv = types.ModuleType('cvxopt')
    cv.solvers = types.SimpleNamespace(options={})
    cv.matrix = lambda *a, **k: None
    sys.modules['cvxopt'] = cv
    base = types.ModuleType('cvxopt.base')

    class _Opaque(object):  # lets pickles that embed cvxopt matrices load; the content is never used
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, state):
            pass

    base.matrix = _Opaque
    base.spmatrix = _Opaque
    sys.modules['cvxopt.base'] = base
    sys.modules['pystan'] = types.ModuleType('pystan')
    sm = types.ModuleType('bayes_drt.stan_models')
    sm.save_pickle = lambda *a, **k: None
    sm.load_pickle = lambda *a, **k: None
    sys.modules['bayes_drt.stan_models'] = sm
    sm2 = types.ModuleType('stan_models')
    sm2.save_pickle = sm.save_pickle
    sm2.load_pickle = sm.load_pickle
    sys.modules['stan_models'] = sm2


_install_stubs()
sys.path.insert(0, REF)
from bayes_drt import matrices as rm  # noqa: E402
from bayes_drt import inversion as ri  # noqa: E402


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-58s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def read_Z(path):
    import pandas as pd
    if path.endswith('.csv'):
        a = pd.read_csv(path)
        return a['Freq'].values.astype(float), a['Zreal'].values + 1j * a['Zimag'].values
    with open(path, 'rb') as fh:
        lines = fh.read().decode('latin-1').splitlines()
    start = [i for i, l in enumerate(lines) if l.startswith('ZCURVE')]
    if start:
        # instrument export (data/experimental/PDAC_*.txt): 'ZCURVE TABLE', a header line naming the tab-separated columns,
        # a units line, then the rows
        cols = lines[start[0] + 1].split('\t')
        rows = [l.split('\t') for l in lines[start[0] + 3:] if l.strip()]
        get = lambda name: np.array([float(r[cols.index(name)]) for r in rows])
        return get('Freq'), get('Zreal') + 1j * get('Zimag')
    a = np.array([[float(v) for v in l.split()] for l in lines if l.strip()])      # three bare columns (DRTtools export)
    return a[:, 0], a[:, 1] + 1j * a[:, 2]


# ---------------------------------------------------------------------------------------- matrices
def gen_matrices():
    cases = {
        # name: (freq, basis_freq or None->tau=None)
        '81x81': (np.logspace(6, -2, 81), None),
        '81x161': (np.logspace(6, -2, 81), np.logspace(10, -6, 161)),
        '41x51': (np.logspace(5, -3, 41)[::-1][::-1], 1 / (2 * np.pi * np.logspace(-2, 3, 51))),
    }
    for name, (f, bf) in cases.items():
        if bf is None:
            tau = None
            tau_l = 1 / (2 * np.pi * f)
        else:
            tau = 1 / (2 * np.pi * bf)
            tau_l = tau
        eps = 1 / np.mean(np.diff(np.log(tau_l)))
        out = dict(freq=f, tau=tau_l, epsilon=eps, tau_is_none=np.array(tau is None))
        for part in ('real', 'imag'):
            out['A_' + part[:2]] = rm.construct_A(f, part, tau=tau, epsilon=eps)
        for o in (0, 1, 2):
            out['L%d' % o] = rm.construct_L(1 / (2 * np.pi * tau_l), tau=tau_l, epsilon=eps, order=o)
            out['M%d' % o] = rm.construct_M(1 / (2 * np.pi * tau_l), order=o, epsilon=eps)
        save('mat_drt_' + name, **out)

    # general (non log-uniform) path, small: irregular frequency grid and irregular tau grid
    rs = np.random.RandomState(7)
    f = np.sort(10 ** rs.uniform(-1, 4, 12))[::-1]
    tau = np.sort(10 ** rs.uniform(-5, 1, 9))
    eps = 1.7
    out = dict(freq=f, tau=tau, epsilon=eps)
    for part in ('real', 'imag'):
        out['A_' + part[:2]] = rm.construct_A(f, part, tau=tau, epsilon=eps)
    for o in (0, 1, 2):
        out['L%d' % o] = rm.construct_L(1 / (2 * np.pi * tau), tau=tau, epsilon=eps, order=o)
        out['M%d' % o] = rm.construct_M(1 / (2 * np.pi * tau), order=o, epsilon=eps)
    # extra construct_L orders the reference supports (matrices.py:278-315)
    out['L3'] = rm.construct_L(1 / (2 * np.pi * tau), tau=tau, epsilon=eps, order=3)
    out['L0p5'] = rm.construct_L(1 / (2 * np.pi * tau), tau=tau, epsilon=eps, order=0.5)
    out['L1p25'] = rm.construct_L(1 / (2 * np.pi * tau), tau=tau, epsilon=eps, order=1.25)
    out['Lmix'] = rm.construct_L(1 / (2 * np.pi * tau), tau=tau, epsilon=eps, order=[0.2, 0.5, 0.3])
    out['Mmix'] = rm.construct_M(1 / (2 * np.pi * tau), order=[0.2, 0.5, 0.3], epsilon=eps)
    save('mat_drt_general_12x9', **out)


def gen_ddt():
    f = np.logspace(6, -2, 81)
    tau = 1 / (2 * np.pi * np.logspace(10, -6, 161))
    eps = 1 / np.mean(np.diff(np.log(tau)))
    out = dict(freq=f, tau=tau, epsilon=eps)
    for bc, sym, tag in (('transmissive', 'planar', 'tp'), ('blocking', 'planar', 'bp'), ('blocking', 'spherical', 'bs')):
        for dt in ('parallel', 'series'):
            for part in ('real', 'imag'):
                A = rm.construct_A(f, part, tau=tau, epsilon=eps, kernel='DDT', dist_type=dt, symmetry=sym, bc=bc)
                out['A_%s_%s_%s' % (part[:2], tag, dt)] = A
    save('ddt_toeplitz_81x161', **out)

    rs = np.random.RandomState(11)
    f = np.sort(10 ** rs.uniform(-2, 5, 7))[::-1]
    tau = np.sort(10 ** rs.uniform(-5, 2, 6))
    eps = 2.3
    out = dict(freq=f, tau=tau, epsilon=eps, k_ct=0.37)
    for bc, sym, tag in (('transmissive', 'planar', 'tp'), ('blocking', 'planar', 'bp'), ('blocking', 'spherical', 'bs')):
        for dt in ('parallel', 'series'):
            for ct in (False, True):
                for part in ('real', 'imag'):
                    A = rm.construct_A(f, part, tau=tau, epsilon=eps, kernel='DDT', dist_type=dt, symmetry=sym,
                                       bc=bc, ct=ct, k_ct=0.37 if ct else None)
                    out['A_%s_%s_%s_ct%d' % (part[:2], tag, dt, int(ct))] = A
    save('ddt_general_7x6', **out)


# ---------------------------------------------------------------------------------------- dat dicts
def _dat_arrays(dat):
    out = {}
    for k, v in dat.items():
        if k.endswith('_tilde'):
            continue  # duplicates of the fit inputs (inversion.py:1744-1746)
        out[k] = np.asarray(v)
    return out


def gen_dat():
    f, Z = read_Z(os.path.join(REF, 'data/simulated/Z_2ZARC_uniform_0.25.csv'))
    for tag, bf in (('K101', None), ('K161', np.logspace(10, -6, 161)), ('K81', 'freq')):
        inv = ri.Inverter(basis_freq=(f if isinstance(bf, str) else bf))
        fs, Zs, WZ_re, WZ_im, W_re, W_im, dist_mat = inv._prep_matrices(f, Z, 'both', weights=None, dZ=False,
                                                                        scale_Z=True, penalty='discrete',
                                                                        fit_type='map')
        for mode in ('optimize', 'sample'):
            dat = inv._prep_stan_data(fs, Zs, 'both', 'Series', dist_mat, False, 0.002, mode=mode,
                                      inductance_scale=1, outlier_lambda=None, fitY=False, SA=False, SASY=False)
            arr = _dat_arrays(dat)
            arr['Z_scale'] = np.array(inv._Z_scale)
            arr['tau'] = inv.distributions['DRT']['tau']
            arr['epsilon'] = np.array(inv.distributions['DRT']['epsilon'])
            arr['freq_in'] = f
            arr['Z_in'] = Z
            save('dat_%s_2ZARC_uniform_0.25_%s' % (mode, tag), **arr)
        if tag == 'K161':
            # outlier variant (package "new" form, inversion.py:1873-1880; N overwritten at :1208-1211)
            for mode in ('optimize', 'sample'):
                dat = inv._prep_stan_data(fs, Zs, 'both', 'Series', dist_mat, True, 0.002, mode=mode,
                                          inductance_scale=1, outlier_lambda=None, fitY=False, SA=False, SASY=False)
                arr = {k: np.asarray(v) for k, v in dat.items() if k.startswith('sigma_out')}
                save('dat_%s_outlier_scalars' % mode, **arr)

    # series + parallel (config 5 family): DRT + TP-DDT, K=161 both
    f, Z = read_Z(os.path.join(REF, 'data/simulated/Z_DRT-2-TpDDT_uniform_0.25.csv'))
    bf = np.logspace(10, -6, 161)
    dists = {'DRT': {'kernel': 'DRT'},
             'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel',
                        'x_scale': 0.8}}
    inv = ri.Inverter(basis_freq=bf, distributions=dists)
    fs, Zs, WZ_re, WZ_im, W_re, W_im, dist_mat = inv._prep_matrices(f, Z, 'both', weights=None, dZ=False,
                                                                    scale_Z=True, penalty='discrete', fit_type='map')
    for mode in ('optimize', 'sample'):
        for outl in (False, True):
            dat = inv._prep_stan_data(fs, Zs, 'both', 'Series-Parallel', dist_mat, outl, 0.002, mode=mode,
                                      inductance_scale=1, outlier_lambda=None, fitY=False, SA=False, SASY=False)
            arr = _dat_arrays(dat)
            if not (mode == 'sample' and not outl):
                # keep only one full copy of the big matrices; the others keep scalars + Z
                arr = {k: v for k, v in arr.items() if v.size < 2000}
            arr['Z_scale'] = np.array(inv._Z_scale)
            arr['freq_in'] = f
            arr['Z_in'] = Z
            save('dat_%s_DRT-TpDDT_%s' % (mode, 'outliers' if outl else 'plain'), **arr)


# ---------------------------------------------------------------------------------------- KATs
def gen_kats():
    sys.path.insert(0, os.path.join(REF, 'code_EchemActa/bayes-drt_20201113'))
    pk = sorted(glob.glob(os.path.join(REF, 'code_EchemActa/map_results/obj_*.pkl')))
    for p in pk:
        name = os.path.basename(p)[4:-4]
        with open(p, 'rb') as fh:
            obj = pickle.load(fh)
        d = obj.__dict__
        out = {}
        res = d['_opt_result']
        for k, v in res.items():
            if k.endswith('_tilde'):
                continue
            out['opt__' + k] = np.asarray(v, dtype=float)
        out['model_name'] = np.array(d['stan_model_name'])
        out['Z_scale'] = np.array(float(d['_Z_scale']))
        out['sigma_min'] = np.array(float(d.get('sigma_min', np.nan)))
        out['f_train'] = np.asarray(d['f_train'], dtype=float)
        dists = d['_distributions']
        names = list(d['distribution_matrices'].keys())
        out['dist_names'] = np.array(names)
        for n in names:
            m = d['distribution_matrices'][n]
            info = dists.get(n, {}) if isinstance(dists, dict) else {}
            for key in ('A_re', 'A_im', 'L0', 'L1', 'L2'):
                if key in m:
                    out['mat__%s__%s' % (n, key)] = np.asarray(m[key], dtype=float)
            out['info__%s__dist_type' % n] = np.array(str(info.get('dist_type', 'series')))
            out['info__%s__kernel' % n] = np.array(str(info.get('kernel', 'DRT')))
            out['info__%s__x_scale' % n] = np.array(float(info.get('x_scale', 1.0)))
            out['info__%s__bc' % n] = np.array(str(info.get('bc', '')))
            out['info__%s__symmetry' % n] = np.array(str(info.get('symmetry', '')))
        # measured data, when the simulated file of the same name exists
        zf = os.path.join(REF, 'data/simulated/Z_%s.csv' % name)
        pdac = 'PDAC_COM3_02109_Contact10_2065C_500C.txt'                   # Run fits.ipynb cells 24, 28
        exp = {'LIB_data_qtr': 'DRTtools_LIB_data_qtr.csv', 'LIB_data_qtr_DRT-TpDDT': 'DRTtools_LIB_data_qtr.csv',
               'LIB_data': 'DRTtools_LIB_data.txt', 'LIB_data_DRT-TpDDT': 'DRTtools_LIB_data.txt',
               'PDAC': pdac, 'PDAC_outliers': pdac, 'PDAC_DRT-TpDDT': pdac, 'PDAC_DRT-TpDDT_outliers': pdac}
        if name in exp:
            zf = os.path.join(REF, 'data/experimental', exp[name])
        if os.path.exists(zf):
            f, Z = read_Z(zf)
            idx = np.argsort(f)[::-1]
            out['data_freq'] = f[idx]
            out['data_Z'] = Z[idx]
        save('kat_' + name, **out)


# ---------------------------------------------------------------------------------------- CSVs + predict
def gen_csv():
    def csv(path):
        import pandas as pd
        a = pd.read_csv(path)
        a = a[[c for c in a.columns if not c.startswith('Unnamed')]]
        return a.values.astype(float), np.array(list(a.columns))

    for stem in ('2ZARC_uniform_0.25', '2ZARC_noiseless', '2RC_uniform_0.25', 'RC-ZARC_uniform_0.25', 'BimodalTP-DDT_uniform_0.25'):
        out = {}
        for key, rel in (('Z', 'data/simulated/Z_%s.csv'), ('Gout_map', 'code_EchemActa/map_results/Gout_%s.csv'),
                         ('Zout_map', 'code_EchemActa/map_results/Zout_%s.csv'),
                         ('Gout_bayes', 'code_EchemActa/bayes_results/Gout_%s.csv'),
                         ('Zout_bayes', 'code_EchemActa/bayes_results/Zout_%s.csv')):
            p = os.path.join(REF, rel % stem)
            if os.path.exists(p):
                out[key], out[key + '_cols'] = csv(p)
        circ = stem.split('_')[0]
        p = os.path.join(REF, 'data/simulated/gamma_%s.csv' % circ)
        if os.path.exists(p):
            out['gamma_true'], out['gamma_true_cols'] = csv(p)
        save('csv_' + stem, **out)
    p = os.path.join(REF, 'code_EchemActa/bayes_results/Gout_2RC_uniform_0.25_4x1000.csv')
    g, c = csv(p)
    save('csv_2RC_uniform_0.25_4x1000', Gout_bayes=g, Gout_bayes_cols=c)


def gen_basis():
    """construct_A with the Cole-Cole and Zic basis functions (matrices.py:14-21; Toeplitz path on a log-uniform grid and the
    general double loop on an irregular one, DRT and one DDT kernel), construct_L for the Zic basis (order 0, :316-318) and
    construct_L on NON-collocated grids (any frequencies against any tau, :268-325)."""
    out = {}
    f_lu = np.logspace(4, -2, 25)
    tau_sup = 1 / (2 * np.pi * np.logspace(5, -3, 33))           # log-uniform superset of 1/(2 pi f): Toeplitz path
    rs = np.random.RandomState(11)
    f_ir = np.sort(10 ** rs.uniform(-2, 4, 9))[::-1]
    tau_ir = np.sort(10 ** rs.uniform(-5, 1, 7))
    out.update(f_lu=f_lu, tau_sup=tau_sup, f_ir=f_ir, tau_ir=tau_ir)
    for basis, eps in (('Cole-Cole', 0.8), ('Zic', 1.0)):
        tag = basis.replace('-', '')
        for part in ('real', 'imag'):
            out['A_%s_toep_%s' % (tag, part)] = rm.construct_A(f_lu, part, tau=tau_sup, basis=basis, epsilon=eps)
            out['A_%s_coll_%s' % (tag, part)] = rm.construct_A(f_lu, part, basis=basis, epsilon=eps)
            out['A_%s_gen_%s' % (tag, part)] = rm.construct_A(f_ir, part, tau=tau_ir, basis=basis, epsilon=eps)
            out['A_%s_ddt_%s' % (tag, part)] = rm.construct_A(f_ir, part, tau=tau_ir, basis=basis, epsilon=eps, kernel='DDT',
                                                            dist_type='parallel', symmetry='planar', bc='transmissive')
        out['eps_' + tag] = np.array(eps)
    out['L_Zic_0'] = rm.construct_L(f_lu, tau=tau_sup, basis='Zic', epsilon=1.0, order=0)
    for order, tag in ((0, '0'), (1, '1'), (2, '2'), (3, '3'), (0.5, 'h'), (1.25, 'q'), ([0.2, 0.5, 0.3], 'mix')):
        out['L_rect_' + tag] = rm.construct_L(f_ir, tau=tau_sup, epsilon=2.5, order=order)
    save('basis_functions', **out)


def gen_hmc_suite():
    """The reference's published HMC study as ONE fixture: the 60 simulated DRT spectra `Run fits.ipynb` cell 3 selects, the
    posterior mean / 2.5 % / 97.5 % curves it stored for each (bayes_results/Gout_*.csv, 2 chains x (200 + 200), cell 5), the
    MAP curves (map_results/Gout_*.csv, cell 4) and the sampler diagnostics pystan printed into the notebook for each run:
    iterations that saturated tree depth 10, divergent iterations, wall time.  Arrays and numbers only."""
    import json
    import re
    import pandas as pd
    nb = json.load(open(os.path.join(REF, 'code_EchemActa', 'Run fits.ipynb')))

    def cell_text(cell):
        out = []
        for o in cell.get('outputs', []):
            t = o.get('text') or o.get('data', {}).get('text/plain') or ''
            out.append(''.join(t))
        return '\n'.join(out)

    def parse(text):
        """file stem -> dict(saturated, divergent, iterations, seconds) from the printed stream of one loop cell."""
        res, cur = {}, None
        for line in text.splitlines():
            m = re.search(r'Z_([^\\/]+)\.csv', line)
            if m:
                cur = m.group(1)
                res[cur] = dict(saturated=0, divergent=0, iterations=0, seconds=np.nan)
                continue
            if cur is None:
                continue
            m = re.search(r'(\d+) of (\d+) iterations saturated the maximum tree depth of (\d+)', line)
            if m:
                res[cur]['saturated'], res[cur]['iterations'] = int(m.group(1)), int(m.group(2))
            m = re.search(r'(\d+) of (\d+) iterations ended with a divergence', line)
            if m:
                res[cur]['divergent'], res[cur]['iterations'] = int(m.group(1)), int(m.group(2))
            m = re.search(r'File fit time: ([0-9.]+) seconds', line)
            if m:
                res[cur]['seconds'] = float(m.group(1))
        return res

    cells = [c for c in nb['cells'] if c['cell_type'] == 'code']
    mcmc = next(c for c in cells if ''.join(c['source']).startswith('"MCMC sampling"'))
    mapc = next(c for c in cells if ''.join(c['source']).startswith('"MAP fits"'))
    four = next(c for c in cells if '4 chains, 1000 iterations' in ''.join(c['source']))
    diag, diag_map = parse(cell_text(mcmc)), parse(cell_text(mapc))
    stems = sorted(diag)
    stems = [s for s in stems if os.path.exists(os.path.join(REF, 'code_EchemActa/bayes_results/Gout_%s.csv' % s))]
    Zs, Gb, Gm, gt, rows = [], [], [], [], []
    for s in stems:
        z = pd.read_csv(os.path.join(REF, 'data/simulated/Z_%s.csv' % s))
        Zs.append(np.array([z['Freq'].values, z['Zreal'].values, z['Zimag'].values]).T)
        gb = pd.read_csv(os.path.join(REF, 'code_EchemActa/bayes_results/Gout_%s.csv' % s))
        Gb.append(gb[['tau', 'gamma', 'gamma_lo', 'gamma_hi']].values.astype(float))
        gm = pd.read_csv(os.path.join(REF, 'code_EchemActa/map_results/Gout_%s.csv' % s))
        Gm.append(gm[['tau', 'gamma']].values.astype(float))
        circ = s.split('_')[0]
        pg = os.path.join(REF, 'data/simulated/gamma_%s.csv' % circ)
        if os.path.exists(pg):                      # RC / 2RC have delta-function distributions: no curve file
            g = pd.read_csv(pg)
            g = g[[c for c in g.columns if not c.startswith('Unnamed')]].values.astype(float)
            gt.append(g[:, :2])
        else:
            gt.append(np.full((200, 2), np.nan))
        d = diag[s]
        rows.append([d['saturated'], d['divergent'], d['iterations'] or 400, d['seconds'], diag_map.get(s, {}).get('seconds', np.nan)])
    t4 = cell_text(four)
    m_sat = re.search(r'(\d+) of (\d+) iterations saturated', t4)
    m_div = re.search(r'(\d+) of (\d+) iterations ended with a divergence', t4)
    m_sec = re.search(r'File fit time: ([0-9.]+) seconds', t4)
    assert len({z.shape for z in Zs}) == 1 and len({g.shape for g in gt}) == 1
    save('hmc_suite', stems=np.array(stems), Z=np.array(Zs), Gout_bayes=np.array(Gb), Gout_map=np.array(Gm),
         gamma_true=np.array(gt), diag=np.array(rows, dtype=float),
         diag_cols=np.array(['saturated_treedepth10', 'divergent', 'iterations', 'hmc_seconds', 'map_seconds']),
         run4x1000=np.array([int(m_sat.group(1)), int(m_div.group(1)), int(m_sat.group(2)), float(m_sec.group(1))]),
         run4x1000_cols=np.array(['saturated_treedepth10', 'divergent', 'iterations', 'seconds']))


def gen_hmc_suite2():
    """The other families of the reference's published HMC study (Run fits.ipynb cells 10, 14, 18, 22 and the stored
    bayes_results/Gout_*.csv): RC-ZARC (Series_pos on its own 51-point basis), trunc (Series, sign-free), BimodalTP-DDT /
    BimodalBP-DDT (ONE parallel diffusion distribution = the `Parallel` model), DRT-k-TpDDT (Series-Parallel_pos) and
    DRT-TpDDT-BpDDT (Series-2Parallel_pos): spectra, stored posterior mean / band curves (every column of the CSV), and the
    diagnostics pystan printed.  Arrays and numbers only; one entry per spectrum, keyed by its file stem."""
    import json
    import re
    import pandas as pd
    nb = json.load(open(os.path.join(REF, 'code_EchemActa', 'Run fits.ipynb')))
    diag = {}
    for c in nb['cells']:
        if c['cell_type'] != 'code' or not ''.join(c['source']).startswith('"MCMC sampling"'):
            continue
        cur = None
        for o in c.get('outputs', []):
            for line in ''.join(o.get('text') or '').splitlines():
                m = re.search(r'Z_([^\\/]+)\.csv', line)
                if m:
                    cur = m.group(1); diag[cur] = [0, 0, 400, np.nan]; continue
                if cur is None:
                    continue
                m = re.search(r'(\d+) of (\d+) iterations saturated', line)
                if m: diag[cur][0] = int(m.group(1)); diag[cur][2] = int(m.group(2))
                m = re.search(r'(\d+) of (\d+) iterations ended with a divergence', line)
                if m: diag[cur][1] = int(m.group(1)); diag[cur][2] = int(m.group(2))
                m = re.search(r'File fit time: ([0-9.]+) seconds', line)
                if m: diag[cur][3] = float(m.group(1))
    out = {}
    stems = []
    for pth in sorted(glob.glob(os.path.join(REF, 'code_EchemActa/bayes_results/Gout_*.csv'))):
        stem = os.path.basename(pth)[5:-4]
        fam = stem.split('_')[0]
        if fam not in ('RC-ZARC', 'trunc', 'BimodalTP-DDT', 'BimodalBP-DDT', 'DRT-2-TpDDT', 'DRT-3-TpDDT', 'DRT-4-TpDDT',
                       'DRT-TpDDT-BpDDT'):
            continue
        zf = os.path.join(REF, 'data/simulated/Z_%s.csv' % stem)
        if not os.path.exists(zf):
            continue
        z = pd.read_csv(zf)
        g = pd.read_csv(pth)
        g = g[[c for c in g.columns if not c.startswith('Unnamed')]]
        key = stem.replace('-', '').replace('.', 'p')
        out['Z__' + key] = np.array([z['Freq'].values, z['Zreal'].values, z['Zimag'].values]).T
        out['G__' + key] = g.values.astype(float)
        out['Gcols__' + key] = np.array(list(g.columns))
        out['diag__' + key] = np.array(diag.get(stem, [np.nan, np.nan, 400, np.nan]), dtype=float)
        stems.append(stem)
    out['stems'] = np.array(stems)
    out['diag_cols'] = np.array(['saturated_treedepth10', 'divergent', 'iterations', 'hmc_seconds'])
    save('hmc_suite2', **out)


def gen_predict():
    # predict_distribution / predict_Z / predict_sigma for a *given* coefficient vector (no Stan needed):
    # inversion.py:3298-3311 (gamma = Phi @ coef), :2942-2959 (Z_hat), :3089-3139 (sigma)
    f, Z = read_Z(os.path.join(REF, 'data/simulated/Z_2ZARC_uniform_0.25.csv'))
    inv = ri.Inverter(basis_freq=np.logspace(10, -6, 161))
    inv._prep_matrices(f, Z, 'both', weights=None, dZ=False, scale_Z=True, penalty='discrete', fit_type='map')
    rs = np.random.RandomState(3)
    K = 161
    tau = inv.distributions['DRT']['tau']
    coef = np.exp(-0.5 * ((np.log(tau) - np.log(1e-2)) / 1.5) ** 2) * (1 + 0.1 * rs.rand(K))
    inv.distribution_fits = {'DRT': {'coef': coef}}
    inv.R_inf = 0.93
    inv.inductance = 2.2e-7
    inv.fit_type = 'map'
    inv.stan_model_name = 'Series_pos_StanModel.pkl'
    inv.error_fit = {'sigma_min': 0.002 * inv._Z_scale, 'sigma_res': 0.004, 'alpha_prop': 0.003, 'alpha_re': 0.002,
                     'alpha_im': 0.001}
    tau_plot = np.logspace(-7, 2, 200)
    g = inv.predict_distribution('DRT', eval_tau=tau_plot)
    f = np.sort(f)[::-1]
    f_alt = f * 0.7  # same length as f_train: the reference cache checks compare element-wise
    Zp = inv.predict_Z(f)
    inv.f_pred = None  # the reference's cache check cannot compare grids of different length
    Zp_alt = inv.predict_Z(f_alt)
    s_re, s_im = inv.predict_sigma(f_alt)
    Rp = inv.predict_Rp()
    save('predict_2ZARC_K161', freq=f, tau=tau, epsilon=np.array(inv.distributions['DRT']['epsilon']), coef=coef,
         R_inf=np.array(inv.R_inf), inductance=np.array(inv.inductance), tau_plot=tau_plot, gamma=g, Z_pred=Zp,
         f_alt=f_alt, Z_pred_alt=Zp_alt, sigma_re_alt=s_re, sigma_im_alt=s_im, Rp=np.array(Rp), Z_scale=np.array(inv._Z_scale),
         **{'err_' + k: np.array(v) for k, v in inv.error_fit.items()})


def gen_host():
    """Host-logic leaves of the Inverter: _format_weights (inversion.py:2338-2395) for every scheme x part, the
    distribution dicts set_distributions produces (:66-127), and _scale_Z's admittance branch (:2417-2434)."""
    from bayes_drt.inversion import Inverter
    f, Z = read_Z(os.path.join(REF, 'data/simulated/Z_2ZARC_uniform_0.25.csv'))
    inv = Inverter()
    out = {'freq': f, 'Z': Z}
    rs = np.random.RandomState(0)
    arr_real = rs.uniform(0.5, 2.0, len(f))
    arr_cplx = rs.uniform(0.5, 2.0, len(f)) + 1j * rs.uniform(0.5, 2.0, len(f))
    cases = {'none': None, 'unity': 'unity', 'modulus': 'modulus', 'Orazem': 'Orazem', 'proportional': 'proportional',
             'prop_adj': 'prop_adj', 'float': 0.7, 'int': 3, 'complex': 0.3 + 1.2j, 'array_real': arr_real,
             'array_complex': arr_cplx}
    out['arr_real'], out['arr_cplx'] = arr_real, arr_cplx
    for name, w in cases.items():
        for part in ('both', 'real', 'imag'):
            try:
                out['w_%s_%s' % (name, part)] = np.asarray(inv._format_weights(f, Z, w, part), dtype=complex)
            except ValueError:
                # numpy >= 1.25: `array == 'unity'` is element-wise and the reference's first test raises for array
                # weights; those cases are checked by property in tests/test_inverter_host.py instead
                pass
    save('host_weights', **out)
    # distribution dicts after the constructor's validation / default filling (values as strings: key=value;...)
    dists = {'DRT': {'kernel': 'DRT'}, 'tp': {'kernel': 'DDT', 'bc': 'transmissive', 'dist_type': 'parallel'},
             'bs': {'kernel': 'DDT', 'symmetry': 'spherical'}, 'ct': {'kernel': 'DDT', 'ct': True, 'k_ct': 2.0, 'dist_type': 'series'}}
    res = {}
    for name, info in dists.items():
        got = Inverter(distributions={name: dict(info)}).distributions[name]
        res[name] = ';'.join('%s=%s' % (k, got[k]) for k in sorted(got))
    save('host_distributions', **{k: np.array(v) for k, v in res.items()})
    # _scale_Z for a single parallel planar DDT (admittance scaling) -- transmissive and blocking targets
    fz, Zz = read_Z(os.path.join(REF, 'data/simulated/Z_BimodalTP-DDT_uniform_0.25.csv'))
    sc = {}
    for bc in ('transmissive', 'blocking'):
        iv = Inverter(distributions={'d': {'kernel': 'DDT', 'dist_type': 'parallel', 'symmetry': 'planar', 'bc': bc}})
        Zs = iv._scale_Z(Zz, 'map')
        sc['scale_' + bc] = np.array(iv._Z_scale)
        sc['Zs_' + bc] = Zs
        Zr = iv._scale_Z(Zz, 'ridge')
        sc['scale_ridge_' + bc] = np.array(iv._Z_scale)
    save('host_scale_parallel', freq=fz, Z=Zz, **sc)
    # fit(part='real' / 'imag') for a two-distribution model: the Stan data keeps N = 2 Nf with the other part zeroed
    # (inversion.py:1892-1905).  Small basis (K = 20, off the frequency grid) to keep the fixture small.
    f2, Z2 = read_Z(os.path.join(REF, 'data/simulated/Z_DRT-2-TpDDT_uniform_0.25.csv'))
    bf = np.logspace(6.3, -2.3, 20)
    dists = {'DRT': {'kernel': 'DRT'},
             'TP-DDT': {'kernel': 'DDT', 'symmetry': 'planar', 'bc': 'transmissive', 'dist_type': 'parallel', 'x_scale': 0.8}}
    out = {'freq': f2, 'Z': Z2, 'basis_freq': bf}
    for part in ('real', 'imag'):
        iv = Inverter(basis_freq=bf, distributions=dists)
        fs, Zs, _, _, _, _, dm = iv._prep_matrices(f2, Z2, part, weights=None, dZ=False, scale_Z=True, penalty='discrete',
                                                   fit_type='map')
        dat = iv._prep_stan_data(fs, Zs, part, 'Series-Parallel', dm, False, 0.002, mode='optimize', inductance_scale=1,
                                 outlier_lambda=None, fitY=False, SA=False, SASY=False)
        for k in ('N', 'Z', 'As', 'Ap'):
            out['%s_%s' % (part, k)] = np.asarray(dat[k])
    save('host_dat_parts', **out)


def gen_ridge():
    """Ridge artefacts of the reference: code_EchemActa/comparisons/hyper-ridge/results, produced by
    `hyper-ridge run fits.ipynb` (cell 4) with the paper snapshot's ridge_fit and cvxopt: Re-Im cross-validation over
    lambda_0 = logspace(-15, 0, 61) (ordinary ridge), then hyper-lambda fits (hl_fbeta) at the best lambda_0.
    Stored per spectrum: the CV curves, the recovered gamma for three f_beta values, and -- from the pickled fit object --
    the matrices of the fit and the complete iteration history: every QP the reference handed to cvxopt (via its
    lambda vector) together with cvxopt's solution and primal objective.  These are known-answer vectors for the QP solver
    that replaces cvxopt."""
    import pandas as pd
    sys.path.append(os.path.join(REF, 'code_EchemActa/bayes-drt_20201113'))
    res = os.path.join(REF, 'code_EchemActa/comparisons/hyper-ridge/results')
    for stem in ('2ZARC_uniform_0.25', '2ZARC_Orazem_0.25', '2ZARC_Macdonald_0.25'):
        f, Z = read_Z(os.path.join(REF, 'data/simulated/Z_%s.csv' % stem))
        out = {'freq': f, 'Z': Z, 'epsilon': np.array(2.0)}
        for fb in ('0.1', '1', '10'):
            g = pd.read_csv(os.path.join(res, 'Gout_%s_fbeta=%s.csv' % (stem, fb)))
            out['tau_plot'] = g['tau'].values
            out['gamma_fbeta_' + fb] = g['gamma'].values
            with open(os.path.join(res, 'obj_%s_fbeta=%s.pkl' % (stem, fb)), 'rb') as fh:
                o = pickle.load(fh)
            d = o.__dict__
            out['coef_fbeta_' + fb] = np.asarray(d['coef_'], dtype=float)
            out['lam2_fbeta_' + fb] = np.asarray(d['lambda_vectors_'][2], dtype=float)
            out['cost_fbeta_' + fb] = np.array(float(d['cost_']))
            out['n_iter_fbeta_' + fb] = np.array(len(d['_iter_history']))
            if fb == '1':
                cv = d['cv_result']
                for k in ('lambda', 'recv', 'imcv', 'totcv'):
                    out['cv_' + k] = cv[k].values.astype(float)
                out['A_re'], out['A_im'] = np.asarray(d['A_re']), np.asarray(d['A_im'])
                out['L2'] = np.asarray(d['L2'])
                out['f_train'] = np.asarray(d['f_train'])
                out['tau'] = np.asarray(d['tau'])
                hist = d['_iter_history']
                pick = sorted(set(i for i in (0, 1, 2, 5, 10, 20, len(hist) - 1) if 0 <= i < len(hist)))
                out['hist_iter'] = np.array(pick)
                out['hist_lam2'] = np.stack([np.asarray(hist[i]['lambda_vectors'][2], dtype=float) for i in pick])
                out['hist_coef'] = np.stack([np.asarray(hist[i]['coef'], dtype=float) for i in pick])
                out['hist_fun'] = np.array([float(hist[i]['fun']) for i in pick])
                out['hist_cost'] = np.array([float(hist[i]['cost']) for i in pick])
                out['hist_status'] = np.array([str(hist[i]['result']['status']) for i in pick])
        save('ridge_' + stem, **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['matrices', 'ddt', 'dat', 'kats', 'csv', 'predict', 'host', 'ridge', 'hmc_suite', 'hmc_suite2', 'basis']
    for w in which:
        globals()['gen_' + w]()
