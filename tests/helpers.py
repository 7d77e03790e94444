"""Shared helpers for the tests: load golden fixtures and turn them into model descriptions."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# L scalings of _prep_stan_data (reference bayes_drt/inversion.py:1725-1737, :1907-1927)
L_SCALE = {
    'optimize': {'series': (0.36, 0.24, 0.12), 'parallel': (0.54, 0.24, 0.12)},
    'sample': {'series': (1.0, 1.0, 0.75), 'parallel': (1.0, 1.0, 0.75)},
}


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def kat_names():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, 'kat_*.npz')))


def rel_l2(a, b):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def kat_to_model(name):
    """Stored Stan `optimizing` result -> (model kwargs, constrained parameter vector, expected outputs).

    The stored fits come from the paper snapshot of the package (code_EchemActa/bayes-drt_20201113); its model
    files equal the package files up to `induc` -> `induc_raw*induc_scale`; the outlier models are
    the old stacked form (sigma_out = 0.05*sigma_out_raw[2Nf] ~ exponential(so_invscale)) = outlier_mode 2.
    """
    d = load('kat_' + name)
    model_name = str(d['model_name'])
    base = model_name.replace('_StanModel.pkl', '')
    fam = base.split('_')[0]
    pos = '_pos' in base
    outl = 'outliers' in base
    names = [str(n) for n in d['dist_names']]
    opt = {k[5:]: d[k] for k in d.files if k.startswith('opt__')}
    dtype = {n: str(d['info__%s__dist_type' % n]) for n in names}
    series = [n for n in names if dtype[n] == 'series']
    par = sorted(n for n in names if dtype[n] == 'parallel')      # sorted-name order (inversion.py:1963-1968)
    order = series + par
    # frequencies actually fitted: the cached matrices can be a superset (SURVEY 8(c) caveat)
    nfit = opt['Z_hat'].shape[0] // 2
    f_train = d['f_train']
    rows = np.arange(len(f_train))
    if 'data_freq' in d.files and len(d['data_freq']) == nfit and len(f_train) != nfit:
        rows = np.array([int(np.argmin(np.abs(np.log(f_train) - np.log(f)))) for f in d['data_freq']])
    elif len(f_train) != nfit:
        rows = None
    if rows is None:
        return None
    blocks = []
    for n in order:
        kind = dtype[n]
        s0, s1, s2 = L_SCALE['optimize'][kind]
        if len(order) == 1:
            s0, s1, s2 = L_SCALE['optimize']['series']     # Series and Parallel models share :1735-1737
        A_re = d['mat__%s__A_re' % n]; A_im = d['mat__%s__A_im' % n]
        A_re = A_re[rows]; A_im = A_im[rows]
        blocks.append(dict(A=np.vstack([A_re, A_im]), L0=s0 * d['mat__%s__L0' % n], L1=s1 * d['mat__%s__L1' % n],
                           L2=s2 * d['mat__%s__L2' % n], parallel=(kind == 'parallel'),
                           nonneg=(pos if kind == 'series' else True), x_scale=float(d['info__%s__x_scale' % n])))
    freq = f_train[rows]
    # parameter vector in Stan declaration order
    if fam == 'Series':
        xkeys, ukeys, dkeys = ['x'], ['ups_raw'], [('d0_strength', 'd1_strength', 'd2_strength')]
    elif fam == 'Series-Parallel':
        xkeys, ukeys = ['xs', 'xp_raw'], ['ups_s_raw', 'ups_p_raw']
        dkeys = [('d0s_strength', 'd1s_strength', 'd2s_strength'), ('d0p_strength', 'd1p_strength', 'd2p_strength')]
    elif fam == 'Series-2Parallel':
        xkeys, ukeys = ['xs', 'xp1_raw', 'xp2_raw'], ['ups_s_raw', 'ups_p1_raw', 'ups_p2_raw']
        dkeys = [tuple('d%d%s_strength' % (i, s) for i in range(3)) for s in ('s', 'p1', 'p2')]
    else:
        return None
    if not all(k in opt for k in xkeys):
        return None                                         # pre-`_raw` snapshot (noiseless 2Parallel)
    # xp_scale is a Stan data item that the pickles do not store (the object's `distributions` can have been
    # edited after the fit); the stored transformed parameter xp = xp_raw*xp_scale pins it.
    for b, key in enumerate(xkeys):
        if key.endswith('_raw') and key[:-4] in opt:
            blocks[b]['x_scale'] = float(np.median(np.asarray(opt[key[:-4]]) / np.asarray(opt[key])))
    induc_raw = opt['induc_raw'] if 'induc_raw' in opt else opt['induc']
    parts = [np.atleast_1d(opt['Rinf_raw']), np.atleast_1d(induc_raw)]
    parts += [opt[k] for k in xkeys]
    parts += [np.atleast_1d(opt[k]) for k in ('sigma_res_raw', 'alpha_prop_raw', 'alpha_re_raw', 'alpha_im_raw')]
    if outl:
        parts.append(opt['sigma_out_raw'])
    parts += [opt[k] for k in ukeys]
    for dk in dkeys:
        parts += [np.atleast_1d(opt[k]) for k in dk]
    params = np.concatenate([np.asarray(p, dtype=float).ravel() for p in parts])
    Z = None
    if 'data_Z' in d.files and len(d['data_Z']) == nfit:
        Zc = d['data_Z'] / float(d['Z_scale'])
        Z = np.concatenate([Zc.real, Zc.imag])
    sigma_min = float(d['sigma_min'])
    if not np.isfinite(sigma_min):
        sigma_min = 0.002
    kw = dict(blocks=blocks, Z=(Z if Z is not None else np.zeros(2 * nfit)), freq=freq, sigma_min=sigma_min,
              ups_alpha=0.05, ups_beta=0.1, induc_scale=1.0, outlier_mode=(2 if outl else 0), so_lambda=10.0,
              use_x_sum=(len(order) > 1), x_sum_invscale=0.0)
    return dict(kw=kw, params=params, opt=opt, has_Z=Z is not None, family=base, xkeys=xkeys)
