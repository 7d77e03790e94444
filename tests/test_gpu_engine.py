"""GPU tests (-m gpu) of the two inference loops: device-resident NUTS vs the recursive CPU oracle NUTS, and the
lock-step L-BFGS fed by GPU evaluations vs the same state machine fed by oracle evaluations."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import load, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_problem():
    """12 frequencies x 9 basis functions (general, non-Toeplitz matrices from the reference): D = 27."""
    m = load('mat_drt_general_12x9')
    A = np.vstack([m['A_re'], m['A_im']])
    blk = dict(A=A, L0=m['L0'], L1=m['L1'], L2=0.75 * m['L2'], nonneg=True)
    rng = np.random.default_rng(0)
    x_true = np.exp(-0.5 * ((np.log(m['tau']) + 4) / 2.0) ** 2)
    Z = A @ x_true + np.concatenate([np.full(12, 0.7), 1e-6 * 2 * np.pi * m['freq']])
    Z = Z / np.std(np.hypot(Z[:12], Z[12:])) + 0.01 * rng.standard_normal(24)
    kw = dict(sigma_min=0.002, ups_alpha=1.0, ups_beta=0.1, induc_scale=1.0)
    return blk, Z, m['freq'], kw


def _bench_problem(mode='sample', tag='K161'):
    d = load('dat_%s_2ZARC_uniform_0.25_%s' % (mode, tag))
    blk = dict(A=d['A'], L0=d['L0'], L1=d['L1'], L2=d['L2'], nonneg=True)
    kw = dict(sigma_min=float(d['sigma_min']), ups_alpha=float(d['ups_alpha']), ups_beta=float(d['ups_beta']),
              induc_scale=float(d['induc_scale']))
    return blk, d['Z'], d['freq'], kw, d


def _ctrl(lib, **kw):
    from bayes_drt_amd._lib import NutsControl
    c = NutsControl(); lib.bdrt_nuts_defaults(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_nuts_matches_oracle_draw_by_draw_small():
    """Same Philox streams, different program structure (checkpointed iteration on the GPU, recursion on the CPU):
    identical decisions, draws equal up to fp64 summation-order noise amplified by the dynamics."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    from oracle import oracle as orc
    blk, Z, f, kw = _small_problem()
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    warm, nd = 10, 8            # short: the dynamics amplify the 1e-13 evaluation noise ~10x per iteration here
    ctrl = _ctrl(prob._lib, max_treedepth=6)
    draws, lp, diag = sample_units(prob, 5, warm, nd, 2024, ctrl)
    for c in range(5):
        ref, lpr, dr = orc.nuts_sample(om, c, 2024, warm, nd, control=orc.nuts_control(max_treedepth=6))
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])      # identical tree shapes
        assert dr['n_divergent'] == diag[c]['n_divergent'] and dr['n_max_treedepth'] == diag[c]['n_max_treedepth']
        assert abs(dr['stepsize'] - diag[c]['stepsize']) < 1e-4 * dr['stepsize']
        err = np.max(np.abs(draws[c] - ref), axis=1) / np.max(np.abs(ref))
        assert np.all(err < 1e-6), err
        assert np.allclose(lp[c], lpr, rtol=1e-6, atol=1e-6)


def test_nuts_matches_oracle_benchmark_shape_short():
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    from oracle import oracle as orc
    blk, Z, f, kw, d = _bench_problem()
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    ctrl = _ctrl(prob._lib, max_treedepth=5)
    draws, lp, diag = sample_units(prob, 2, 6, 4, 1234, ctrl)
    for c in range(2):
        ref, lpr, dr = orc.nuts_sample(om, c, 1234, 6, 4, control=orc.nuts_control(max_treedepth=5))
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog']
        assert np.max(np.abs(draws[c] - ref)) < 1e-6 * np.max(np.abs(ref))
        assert np.allclose(lp[c], lpr, rtol=1e-8, atol=1e-6)


def test_nuts_is_independent_of_packing_and_launch_slicing():
    """Draws are a function of (seed, chain id) only: not of which workgroup / column a chain lands in, nor of
    how the run is cut into kernel launches."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    blk, Z, f, kw = _small_problem()
    prob = Problem([blk], Z, f, **kw)
    ctrl = _ctrl(prob._lib, max_treedepth=5)
    ids = np.arange(20, dtype=np.int32)
    full, _, _ = sample_units(prob, 20, 20, 10, 7, ctrl, chain_ids=ids)
    part, _, _ = sample_units(prob, 3, 20, 10, 7, ctrl, chain_ids=ids[[17, 2, 9]])
    assert np.array_equal(part, full[[17, 2, 9]])
    sliced, _, _ = sample_units(prob, 3, 20, 10, 7, ctrl, chain_ids=ids[[17, 2, 9]], rounds_per_launch=7)
    assert np.array_equal(sliced, part)
    other, _, _ = sample_units(prob, 3, 20, 10, 8, ctrl, chain_ids=ids[[17, 2, 9]])
    assert not np.array_equal(other, part)
    # few chains are spread one per workgroup / wave (the reference's own call shape is 2-4 chains); forcing other
    # packings -- all 20 chains on two workgroups, three per workgroup -- gives the same bits
    try:
        for cpw in ('16', '3', '1'):
            os.environ['BDRT_CHAINS_PER_WG'] = cpw
            packed, _, _ = sample_units(prob, 20, 20, 10, 7, ctrl, chain_ids=ids)
            assert np.array_equal(packed, full), cpw
    finally:
        os.environ.pop('BDRT_CHAINS_PER_WG', None)


def test_few_chain_packing_benchmark_shape_bitwise():
    """81 x 161 (the LDS-resident fast path): 4 chains on 4 workgroups (default) == the same 4 chains packed in one."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    blk, Z, f, kw, d = _bench_problem()
    prob = Problem([blk], Z, f, **kw)
    ctrl = _ctrl(prob._lib, max_treedepth=6)
    spread, lp0, dg0 = sample_units(prob, 4, 12, 6, 1234, ctrl)
    try:
        os.environ['BDRT_CHAINS_PER_WG'] = '16'
        packed, lp1, dg1 = sample_units(prob, 4, 12, 6, 1234, ctrl)
    finally:
        os.environ.pop('BDRT_CHAINS_PER_WG', None)
    assert np.array_equal(spread, packed) and np.array_equal(lp0, lp1)
    assert [x['n_leapfrog'] for x in dg0] == [x['n_leapfrog'] for x in dg1]


def test_nuts_posterior_matches_oracle_statistically():
    """Longer runs: GPU chains and oracle chains (different chain ids => independent streams) agree on the posterior
    mean of every parameter within Monte-Carlo error."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    from oracle import oracle as orc
    blk, Z, f, kw = _small_problem()
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    warm, nd = 300, 400
    g, _, dg = sample_units(prob, 16, warm, nd, 5, chain_ids=np.arange(100, 116, dtype=np.int32))
    o = np.stack([orc.nuts_sample(om, c, 5, warm, nd)[0] for c in range(4)])
    gm, om_ = g.reshape(-1, prob.D), o.reshape(-1, prob.D)
    sd = om_.std(axis=0)
    # conservative effective sample sizes: 1/10 of the draws
    se = sd * np.sqrt(1.0 / (gm.shape[0] / 10) + 1.0 / (om_.shape[0] / 10))
    z = np.abs(gm.mean(axis=0) - om_.mean(axis=0)) / se
    assert np.max(z) < 5.0, (np.argmax(z), np.max(z))
    assert np.all(np.abs(gm.std(axis=0) / sd - 1) < 0.25)
    assert np.mean([d['mean_accept'] for d in dg]) > 0.7


def _harness():
    so = os.path.join(ROOT, 'tests', 'host', 'liblbfgs_oracle.so')
    src = os.path.join(ROOT, 'tests', 'host', 'lbfgs_oracle.cpp')
    from oracle import oracle as orc
    orc.build()
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', src, '-L' + os.path.join(ROOT, 'oracle'),
                               '-loracle', '-Wl,-rpath,' + os.path.join(ROOT, 'oracle'), '-o', so])
    return C.CDLL(so)


def _gamma(d, x):
    tau_plot = np.logspace(-7, 2, 200)
    eps = float(d['epsilon'])
    Phi = np.exp(-(eps * np.log(tau_plot[:, None] / d['tau'][None, :])) ** 2)
    return Phi @ x


@pytest.mark.parametrize('tag', ['K81', 'K161'])
def test_map_gpu_vs_oracle(tag):
    """BASELINE config 2: the MAP computed with GPU evaluations equals the MAP computed with CPU-oracle evaluations
    (same L-BFGS + Newton state machines): gamma(ln tau) within 1e-4 rel-L2 -- the tolerance north_star states.
    (The L-BFGS iterate paths diverge after ~100 iterations -- chaotic, SURVEY H1 -- but both runs end at the same
    stationary point because the Newton phase converges to |grad|_inf < 1e-8.)"""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import optimize_batch
    from oracle import oracle as orc
    blk, Z, f, kw, d = _bench_problem('optimize', tag)
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    th0 = np.random.RandomState(1234).uniform(-2, 2, prob.D)
    out, rep = optimize_batch(prob, th0[None])
    assert rep[0]['return_code'] == 0 and rep[0]['grad_inf'] < 1e-8, rep[0]
    h = _harness()
    ref = np.empty(prob.D); ni = C.c_int(); lp = C.c_double(); gi = C.c_double()
    rc = h.harness_optimize_newton(C.byref(om.m), th0.ctypes.data_as(C.c_void_p), 1000, 2000, C.c_double(1e-8),
                                   ref.ctypes.data_as(C.c_void_p), C.byref(ni), C.byref(lp), C.byref(gi))
    assert rc == 0
    K = prob.Ks[0]
    err = rel_l2(_gamma(d, np.exp(out[0][2:2 + K])), _gamma(d, np.exp(ref[2:2 + K])))
    assert abs(rep[0]['lp'] - lp.value) < 1e-8 * abs(lp.value), (rep[0]['lp'], lp.value)
    assert err < 1e-4, err
    # a different start reaches the same MAP
    out2, rep2 = optimize_batch(prob, np.random.RandomState(7).uniform(-2, 2, prob.D)[None])
    assert rel_l2(_gamma(d, np.exp(out2[0][2:2 + K])), _gamma(d, np.exp(out[0][2:2 + K]))) < 1e-4
    # and it is (much) better than the Stan-style early-terminated L-BFGS iterate
    out3, rep3 = optimize_batch(prob, th0[None], max_iter=50000, newton_max_iter=0)
    assert rep[0]['lp'] > rep3[0]['lp']


def _usable_kats():
    from tests.helpers import kat_names, kat_to_model
    out = []
    for n in kat_names():
        k = kat_to_model(n)
        if k is not None and k['has_Z']:
            out.append(n)
    return out


# distance between the stored coefficients (an L-BFGS iterate that stopped by a tolerance test) and the
# stationary point reached from them, per model family: measured maxima (profiles/r03/map_kats.txt) with head-room.  The
# sign-free `Series` fits of the truncated spectra sit in long flat valleys (stored gradient max-norm up to 10): there the
# iterate is far from the optimum in coefficient space while both reproduce the spectrum.  For `Series` / `Series_outliers`
# (bound 4.6 = 460 %) the coefficient distance is therefore REPORT-ONLY: what the test asserts for them is lp >= lp_stored,
# the predicted spectrum (Z_hat) and -- identifiable in the flat valleys -- gamma on the well-determined directions (below).
_COEF_BOUND = {'Series_pos': 0.75, 'Series-Parallel_pos': 1.0, 'Series-2Parallel_pos': 0.4, 'Series': 4.6,
               'Series_outliers': 4.6, 'Series-Parallel_pos_outliers': 1.0}


_PROJ_BOUND = 0.5      # (measured over the 36 fits, profiles/r04/map_kats.txt: median 1.4e-2, maximum 0.36)


@pytest.mark.parametrize('name', _usable_kats())
def test_map_vs_reference_stored_fit(name):
    """Against EVERY usable stored Stan MAP of the reference (code_EchemActa/map_results/obj_*.pkl, 36 of 37; all model families
    incl. Series-Parallel_pos_outliers): started from the stored point, our MAP reaches a log-posterior at least as high,
    evaluated with the same density (SURVEY H1 ladder (c)); the impedance it predicts stays on the stored one; the distance in
    coefficient space is reported and bounded per family, not asserted to 1e-4 (the stored point is an un-converged iterate)."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import optimize_batch
    from tests.helpers import kat_to_model
    k = kat_to_model(name)
    prob = Problem(**k['kw'])
    lay = prob.layout()
    th_ref = prob.unconstrain(k['params'])
    lp_ref, g_ref = prob.logp_grad(th_ref[None], jacobian=False)
    out, rep = optimize_batch(prob, th_ref[None])
    assert rep[0]['lp'] >= lp_ref[0] - 1e-9 * max(1.0, abs(lp_ref[0]))
    assert rep[0]['return_code'] in (0, 2), rep[0]             # 2: the optimum presses against the x_sum >= 0 wall
    con = prob.constrain(out)

    def coef(p):
        return np.concatenate([p[lay['x'][b]:lay['x'][b] + K] for b, K in enumerate(prob.Ks)])
    d = rel_l2(coef(con[0]), coef(k['params']))
    _, Zh, _ = prob.transformed(out)
    dz = rel_l2(Zh[0], k['opt']['Z_hat'])
    # identifiable part of the coefficients: their projection on the well-determined right singular directions of the first
    # block's A (singular value >= 1 % of the largest) -- what the data pin down even in the flat valleys
    A0 = np.asarray(k['kw']['blocks'][0]['A'], dtype=float)
    K0 = prob.Ks[0]
    U, sv, Vt = np.linalg.svd(A0, full_matrices=False)
    V = Vt[sv >= 1e-2 * sv[0]]
    x_ours, x_ref = con[0][lay['x'][0]:lay['x'][0] + K0], k['params'][lay['x'][0]:lay['x'][0] + K0]
    dproj = float(np.linalg.norm(V @ (x_ours - x_ref)) / np.linalg.norm(V @ x_ref))
    print('%s [%s]: lp stored %.4f -> %.4f, |g stored|inf %.3e, coef rel-L2 %.3e (well-determined directions: %.3e, %d of %d), '
          'Z_hat rel-L2 %.3e, rc %d' % (name, k['family'], lp_ref[0], rep[0]['lp'], np.max(np.abs(g_ref)), d, dproj, len(V), K0, dz,
                                        rep[0]['return_code']))
    assert d < _COEF_BOUND[k['family']], (k['family'], d)
    assert dproj < _PROJ_BOUND, (k['family'], dproj)
    assert dz < 0.05, dz


def test_stan_lbfgs_started_at_the_stored_iterates_stops_there():
    """Pin of Stan's L-BFGS TERMINATION (reference call site bayes_drt/inversion.py:1216): each stored Stan MAP is the iterate at
    which one of Stan's tolerance tests fired.  The Stan-style L-BFGS (newton_max_iter = 0) started AT that point must stop by
    a tolerance test as well -- it has no history, so where the stored point sits in a flat valley its first steepest-descent
    steps can still find a decrease that Stan's last quasi-Newton step did not; what is asserted is the distribution
    (profiles/r04/lbfgs_pin.txt: 25 of 36 stop after ONE iteration having moved the coefficients by 1e-9 ... 1e-5, all 36 stop by a
    tolerance test, the 11 that walk on move the spectrum by <= 1.3e-3 and the well-determined coefficient directions by <= 1.2 %)."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import optimize_batch
    from tests.helpers import kat_to_model
    rows = []
    for name in _usable_kats():
        k = kat_to_model(name)
        prob = Problem(**k['kw'])
        lay = prob.layout()
        th = prob.unconstrain(k['params'])
        lp0, _ = prob.logp_grad(th[None], jacobian=False)
        out, rep = optimize_batch(prob, th[None], newton_max_iter=0)
        con = prob.constrain(out)
        K0 = prob.Ks[0]
        A0 = np.asarray(k['kw']['blocks'][0]['A'], dtype=float)
        _, sv, Vt = np.linalg.svd(A0, full_matrices=False)
        V = Vt[sv >= 1e-2 * sv[0]]
        xo, xr = con[0][lay['x'][0]:lay['x'][0] + K0], k['params'][lay['x'][0]:lay['x'][0] + K0]
        dproj = float(np.linalg.norm(V @ (xo - xr)) / np.linalg.norm(V @ xr))
        _, Zh, _ = prob.transformed(out)
        dz = rel_l2(Zh[0], k['opt']['Z_hat'])
        rows.append((name, rep[0]['iterations'], rep[0]['return_code'], rep[0]['lp'] - lp0[0], dproj, dz))
        prob.close()
        assert rep[0]['return_code'] == 0, (name, rep[0])                 # a tolerance test, not the iteration cap
        assert rep[0]['lp'] >= lp0[0] - 1e-9 * max(1.0, abs(lp0[0]))
        assert dproj <= 3e-2 and dz <= 3e-3, (name, dproj, dz)
    its = np.array([r[1] for r in rows]); mv = np.array([r[4] for r in rows])
    at_once = (its <= 3) & (mv <= 1e-4)
    print('stops at the stored iterate (<= 3 iterations, coefficients moved <= 1e-4): %d of %d; iterations median %d max %d' % (
        at_once.sum(), len(rows), np.median(its), its.max()))
    assert at_once.sum() >= 0.6 * len(rows), [(r[0], r[1], r[4]) for r in rows if not (r[1] <= 3 and r[4] <= 1e-4)]
    assert its.max() <= 30000


def test_batched_optimize_equals_single_fits():
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import optimize_batch
    blk, Z, f, kw, d = _bench_problem('optimize', 'K81')
    rng = np.random.default_rng(3)
    Zs = np.stack([Z, Z * 1.02 + 0.002 * rng.standard_normal(len(Z)), Z * 0.97])
    prob = Problem([blk], Zs, f, **kw)
    th0 = np.random.RandomState(5).uniform(-2, 2, (3, prob.D))
    both, rb = optimize_batch(prob, th0, spec=[0, 1, 2], max_iter=300, newton_max_iter=15)
    for i in range(3):
        one, r1 = optimize_batch(prob, th0[i][None], spec=[i], max_iter=300, newton_max_iter=15)
        assert np.array_equal(one[0], both[i]) and r1[0]['iterations'] == rb[i]['iterations']


@pytest.mark.parametrize('n_spectra', [96, 512])
def test_batched_map_of_many_spectra_equals_single_fits(n_spectra):
    """A batch of spectra (512 = BASELINE config 4's spectrum count) through the whole MAP pipeline (lock-step L-BFGS + the
    device-resident Newton polish, one workgroup per fit): every fit converges, and picks of the batch are bit-identical to
    the same fits run alone."""
    import bench
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import optimize_batch
    blk, Z, f, kw, d = _bench_problem('optimize', 'K81')
    fs, Zs = bench.synth_spectra(n_spectra)
    order = np.argsort(fs)[::-1]
    assert np.allclose(fs[order], f)                       # same 81-point grid as the fixture's matrices
    prob = Problem([blk], Zs, f, **kw)
    th0 = np.random.RandomState(9).uniform(-2, 2, (n_spectra, prob.D))
    allx, rep = optimize_batch(prob, th0, spec=np.arange(n_spectra), max_iter=400, lbfgs_before_newton=400)
    assert all(r['return_code'] == 0 and r['grad_inf'] < 1e-8 for r in rep), [r['return_code'] for r in rep]
    for i in (0, n_spectra // 2 - 7, n_spectra - 1):
        one, r1 = optimize_batch(prob, th0[i][None], spec=[i], max_iter=400, lbfgs_before_newton=400)
        assert np.array_equal(one[0], allx[i]) and r1[0]['newton_iterations'] == rep[i]['newton_iterations']


@pytest.mark.parametrize('kernel', ['one_chain_per_workgroup', 'sixteen_chains'])
@pytest.mark.parametrize('family', ['series_outliers_K161', 'series_parallel_2block', 'series_K192_Nf96', 'series_K161_Nf107'])
def test_nuts_matches_oracle_on_the_wide_parameter_vectors(family, kernel, monkeypatch):
    """The sampler kernels for 352 < D <= 512 and D > 512 (outlier error model: D = 493; two blocks of 161: D = 656) against the
    recursive oracle: identical tree shapes, draws equal to summation-order noise (short runs).  Three chains would take the
    one-chain-per-workgroup kernels (bdrt_solo.h / bdrt_solo_wide.h); `sixteen_chains` keeps them on the 16-chain kernel."""
    if kernel == 'sixteen_chains':
        monkeypatch.setenv('BDRT_WIDE1', '0'); monkeypatch.setenv('BDRT_SOLO', '0')
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units
    from oracle import oracle as orc
    from tests.helpers import kat_to_model
    if family == 'series_outliers_K161':
        blk, Z, f, kw, d = _bench_problem('sample', 'K161')
        so = load('dat_sample_outlier_scalars')
        kw = dict(kw, outlier_mode=1, so_lambda=float(so['sigma_out_lambda']), so_alpha=float(so['sigma_out_alpha']),
                  so_beta=float(so['sigma_out_beta']))
        args = dict(blocks=[blk], Z=Z, freq=f, **kw)
    elif family == 'series_K161_Nf107':         # 107 frequencies: the fourth frequency slot per lane, theta rows in LDS
        from tests.test_gpu_edges import _problem
        blk, Z, f, kw = _problem(107, 161)
        args = dict(blocks=[blk], Z=Z, freq=f, **kw)
    elif family == 'series_K192_Nf96':          # largest problem of the S1 evaluator: sampler state in HBM (D = 393)
        from tests.test_gpu_edges import _problem
        blk, Z, f, kw = _problem(96, 192)
        args = dict(blocks=[blk], Z=Z, freq=f, **kw)
    else:
        from bayes_drt_amd.engine import blocks_from_dat
        dd = load('dat_sample_DRT-TpDDT_plain')
        blocks, kw2, _ = blocks_from_dat('Series-Parallel_pos_StanModel.pkl', {k: dd[k] for k in dd.files})
        args = dict(blocks=blocks, Z=dd['Z'], freq=dd['freq'], **kw2)
    prob = Problem(**args)
    om = orc.OracleModel(**args)
    assert prob.D > 352 or family == 'series_K161_Nf107'
    warm, nd, n_units = 6, 4, 3
    ctrl = _ctrl(prob._lib, max_treedepth=5)
    draws, lp, diag = sample_units(prob, n_units, warm, nd, 99, ctrl)
    for c in range(n_units):
        ref, lpr, dr = orc.nuts_sample(om, c, 99, warm, nd, control=orc.nuts_control(max_treedepth=5))
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])
        assert dr['n_divergent'] == diag[c]['n_divergent']
        err = np.max(np.abs(draws[c] - ref), axis=1) / np.max(np.abs(ref))
        assert np.all(err < 1e-6), err


@pytest.mark.parametrize('tile,seed', [('half_wave', 13), ('half_wave', 11), ('generic', 13), ('one_chain', 13), ('one_chain', 11)])
def test_wide_path_long_run_matches_oracle(tile, seed, monkeypatch):
    """The wide-vector sampler path (D = 656: the chain's own pass + the cooperative phase of bdrt_nuts_wide.h) through
    everything a real run meets: trees up to depth 7 (leaves that merge more than four levels go to the cooperative phase),
    subtree closes in both directions, divergent transitions (seed 11), transition ends inside a metric-adaptation window
    (Welford update, window end with a new metric, step-size search restarted) and sampling draws -- draw by draw against
    the recursive oracle.  (Seeds on which no decision of the 46 transitions sits within rounding noise of its threshold:
    kernel and oracle sum in different orders, and a chain that flips one decision is a different chain from there on.)"""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import sample_units, blocks_from_dat
    from oracle import oracle as orc
    if tile == 'generic': monkeypatch.setenv('BDRT_GENERIC_TILE', '1')
    if tile != 'one_chain': monkeypatch.setenv('BDRT_WIDE1', '0')       # three chains: else the kernel of bdrt_solo_wide.h
    dd = load('dat_sample_DRT-TpDDT_plain')
    blocks, kw2, _ = blocks_from_dat('Series-Parallel_pos_StanModel.pkl', {k: dd[k] for k in dd.files})
    args = dict(blocks=blocks, Z=dd['Z'], freq=dd['freq'], **kw2)
    prob = Problem(**args)
    om = orc.OracleModel(**args)
    assert prob.D > 512
    warm, nd, n_units = 40, 6, 3
    ctrl = _ctrl(prob._lib, max_treedepth=7)
    draws, lp, diag = sample_units(prob, n_units, warm, nd, seed, ctrl)
    deep = 0
    for c in range(n_units):
        ref, lpr, dr = orc.nuts_sample(om, c, seed, warm, nd, control=orc.nuts_control(max_treedepth=7))
        assert dr['n_leapfrog'] == diag[c]['n_leapfrog'], (c, dr, diag[c])
        assert dr['n_divergent'] == diag[c]['n_divergent']
        assert abs(dr['stepsize'] - diag[c]['stepsize']) < 1e-4 * dr['stepsize']
        err = np.max(np.abs(draws[c] - ref), axis=1) / np.max(np.abs(ref))
        assert np.all(err < 1e-5), err
        deep += diag[c]['n_leapfrog'] > 40 * nd
    assert deep > 0 or seed == 11                          # trees deeper than 5 doublings were built


@pytest.mark.parametrize('family', ['headline', 'DRT-2-TpDDT_uniform_0.25', 'PDAC_DRT-TpDDT_outliers'])
def test_device_resident_lbfgs_takes_the_host_state_machines_decisions(monkeypatch, family):
    """algorithm='LBFGS' (Stan 2.19's L-BFGS restated, bdrt_lbfgs.h) runs as ONE kernel, a workgroup per fit
    (bdrt_lbfgs_dev.h), with the decisions of the host state machine: over the first iterations the two paths -- device
    reductions vs sequential host sums of the same products -- stay together to rounding; the complete run improves on them and
    ends by one of Stan's tolerance tests or at Stan's iteration cap."""
    from bayes_drt_amd.model import Problem
    from bayes_drt_amd.engine import optimize_batch
    from tests.helpers import kat_to_model
    if family == 'headline':
        blk, Z, f, kw, d = _bench_problem('optimize', 'K81')
        prob = Problem([blk], Z, f, **kw)
        th0 = np.random.RandomState(1234).uniform(-2, 2, (2, prob.D))
    else:
        k = kat_to_model(family)
        prob = Problem(**k['kw'])
        th0 = prob.unconstrain(k['params'])[None] + 0.3 * np.random.RandomState(5).standard_normal((2, prob.D))
    dev, rd = optimize_batch(prob, th0, max_iter=25, newton_max_iter=0)
    monkeypatch.setenv('BDRT_HOST_LBFGS', '1')
    host, rh = optimize_batch(prob, th0, max_iter=25, newton_max_iter=0)
    monkeypatch.delenv('BDRT_HOST_LBFGS')
    for i in range(2):
        assert rd[i]['iterations'] == rh[i]['iterations'] == 25 and rd[i]['return_code'] == rh[i]['return_code'] == 1
        assert rd[i]['n_evals'] == rh[i]['n_evals'], (rd[i], rh[i])                  # the same line-search decisions
        assert abs(rd[i]['lp'] - rh[i]['lp']) <= 1e-7 * max(1.0, abs(rh[i]['lp'])), (rd[i]['lp'], rh[i]['lp'])
        assert np.max(np.abs(dev[i] - host[i])) <= 1e-6 * max(1.0, np.max(np.abs(host[i])))
    full, rf = optimize_batch(prob, th0[:1], max_iter=50000, newton_max_iter=0)
    print('%s: L-BFGS alone ends after %d iterations (%d evaluations), rc %d, lp %.4f, |g|inf %.2e'
          % (family, rf[0]['iterations'], rf[0]['n_evals'], rf[0]['return_code'], rf[0]['lp'], rf[0]['grad_inf']))
    # It ends by one of Stan's tolerance tests (0, -2) or at Stan's own iteration cap (1; the reference's fits set iter = 50000 and
    # take 1.2-2.9 s, i.e. they run to the neighbourhood of the cap as well): which of the two depends on the rounding of the evaluator
    # -- the path is chaotic (profiles/r03/map_timing.txt: 4884, 6020 and 50000 iterations on three neighbouring problems).
    assert rf[0]['return_code'] in (0, -2, 1) and 50 < rf[0]['iterations'] <= 50000, rf[0]
    assert rf[0]['lp'] > max(rd[0]['lp'], rh[0]['lp'])
