"""GPU tests (-m gpu): a fixed slice of the randomised parity cases of tests/fuzz_parity.py (matrices, both evaluators, a short
NUTS run and the MAP of a random problem against the oracle).  The soak over cases 0 ... 599 is recorded in
profiles/r02/fuzz_parity.txt; it found the two limits pinned below."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# the first 24 cases + the extremes of the generator: 25 (K = 192, Nf = 128), 7 / 287 (two distributions of 129 at 128 / 107
# frequencies, with the stacked outlier model: D = 742), 3 (K = 6 ... 8), 16 (6 frequencies, 215 basis functions), 598
# ... and four cases on the shapes of the Toeplitz-table evaluator (every tenth case from 1000 on)
CASES = sorted(set(range(24)) | {25, 287, 457, 598, 3017, 3027, 3047, 3057})


@pytest.mark.parametrize('n', CASES)
def test_random_problem_matches_the_oracle(n):
    from tests.fuzz_parity import run_case
    status, text = run_case(n)
    # (case 14 -- three distributions of 129 at 107 frequencies, beyond the LDS budget of the tiles -- takes the streamed
    #  evaluator of bdrt_big.h since round 4: every stage against the oracle like the others)
    assert status == 'ok', text


def test_problems_beyond_the_lds_budget_take_the_streamed_path():
    """Cases 207 and 283 (three distributions, > 107 frequencies) need 155-160 KiB of LDS for the tile evaluator alone: refused
    at problem creation until round 3, they are evaluated, optimised and sampled by the streamed path of bdrt_big.h now
    (evaluator code 5) -- every stage of the case against the oracle."""
    from bayes_drt_amd.model import Problem
    from tests.fuzz_parity import make_case, run_case
    for n in (207, 283):
        case, text = make_case(n)
        prob = Problem(case['blocks'], case['Z'], case['freq'], **case['kw'])
        assert prob.evaluator() == 5
        prob.close()
        status, text = run_case(n)
        assert status == 'ok', text


def test_parameter_vectors_beyond_864_are_refused_by_the_sampler_not_mis_sampled():
    """Three distributions + the stacked outlier model (no such family in the reference) can exceed the sampler's 27 elements per
    lane; evaluation and MAP work, bdrt_sampler_create says so."""
    from tests.fuzz_parity import run_case
    status, text = run_case(403)
    assert status == 'skip' and 'not supported' in text, text


@pytest.mark.parametrize('first', [0, 20, 40, 350])
def test_random_ridge_fits_device_loop_equals_host_loop(first):
    """tests/fuzz_ridge.py, 20 cases per test (record of cases 0 ... 999: profiles/r02/fuzz_ridge.txt)."""
    from tests.fuzz_ridge import run_case
    for n in range(first, first + 20):
        status, text = run_case(n)
        assert status == 'ok', 'case %d %s' % (n, text)


@pytest.mark.parametrize('first', [0, 20, 40, 200, 300, 340, 360])
def test_random_spectra_through_inverter_fit(first):
    """tests/fuzz_inverter.py, 20 cases per test (record of cases 0 ... 399: profiles/r02/fuzz_inverter.txt)."""
    from tests.fuzz_inverter import run_case
    for n in range(first, first + 20):
        status, text = run_case(n)
        assert status == 'ok', 'case %d %s' % (n, text)


@pytest.mark.parametrize('n', [178, 213, 251])
def test_map_keeps_the_better_of_the_random_and_the_ridge_start(n, monkeypatch):
    """Spectra (128 / 101 / 128 points; unconstrained coefficients or the outlier model) on which the random start alone ends in
    a poor local maximum of the posterior -- everything explained as noise, or a huge Z_hat with a proportionally huge error:
    with the ridge start in the same batch the fit is the (much) higher maximum, and it follows the spectrum."""
    import warnings
    from bayes_drt_amd.inversion import Inverter
    from tests.fuzz_inverter import make_case
    case, _ = make_case(n)
    res = {}
    for single in (True, False):
        if single:
            monkeypatch.setenv('BDRT_MAP_SINGLE_START', '1')
        else:
            monkeypatch.delenv('BDRT_MAP_SINGLE_START')
        inv = Inverter(basis_freq=case['bf'])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            inv.fit(case['f'], case['Z'], **case['kw'])
        rms = np.sqrt(np.mean(np.abs(inv.predict_Z(case['f']) - case['Z']) ** 2))
        res[single] = (inv._opt_report, rms)
    (r1, rms1), (r2, rms2) = res[True], res[False]
    assert 'starts' not in r1 or len(r1['starts']) == 1
    assert r2['start'] == 1 and len(r2['starts']) == 2 and r2['return_code'] == 0
    assert r2['lp'] > r1['lp'] + 100 and rms2 < 0.5 * rms1, (r1['lp'], r2['lp'], rms1, rms2)


def test_random_percentile_and_summary_cases_against_numpy():
    """tests/fuzz_post.py, cases 0 ... 59 (record of 400: profiles/r02/fuzz_post.txt)."""
    from tests.fuzz_post import run_case
    for n in range(60):
        status, text = run_case(n)
        assert status == 'ok', 'case %d %s' % (n, text)
