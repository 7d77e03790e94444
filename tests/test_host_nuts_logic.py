"""Host logic (no GPU): the adaptation bookkeeping the three sampler kernels share (bdrt_nuts_device.h: end of a transition,
dual averaging, metric windows) compiled for the CPU (tests/host/nuts_logic_harness.cpp) against Stan 2.19's documented
schedule and a direct evaluation of the dual-averaging recurrences in numpy."""
import math
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, warmup, n_draws=5):
    exe = str(tmp_path / 'nuts_logic_harness')
    subprocess.check_call(['g++', '-O1', '-std=c++17', os.path.join(ROOT, 'tests/host/nuts_logic_harness.cpp'), '-o', exe])
    rows, head, tot = [], None, None
    for line in subprocess.check_output([exe, str(warmup), str(n_draws)]).decode().strip().splitlines():
        t = line.split()
        if t[0] == 'buffers': head = [int(v) for v in t[1:]]
        elif t[0] == 'totals': tot = dict(n_post=int(t[2]), n_div=int(t[4]), n_leap_total=int(t[6]))
        else: rows.append(dict(it=int(t[1]), next=int(t[3]), draw=int(t[5]), welf=int(t[7]), wend=int(t[9]), wn=float(t[11]),
                               eps=float(t[13]), xbar=float(t[15]), counter=int(t[17]), phase=int(t[19])))
    return head, rows, tot


@pytest.mark.parametrize('warmup,buffers,ends', [
    (1000, (75, 50, 25), [99, 149, 249, 449, 949]),          # Stan's defaults: windows of 25, 50, 100, 200, then stretched to 500
    (150, (75, 50, 25), [99]),                                # 75 + 25 + 50 just fits: one window, stretched to the end
    (140, (21, 14, 105), [125]),                              # it does not fit: 15 % / 75 % / 10 %
    (40, (6, 4, 30), [35]),
    (19, (19, 0, 0), []),                                     # fewer than 20 warm-up iterations: no metric adaptation at all
])
def test_window_schedule_is_stans(tmp_path, warmup, buffers, ends):
    head, rows, tot = _run(tmp_path, warmup)
    assert tuple(head) == buffers
    assert [r['it'] for r in rows if r['wend']] == ends
    init, term = buffers[0], buffers[1]
    for r in rows:
        # Welford samples are taken inside the windows only; the count restarts after every window
        assert r['welf'] == int(bool(ends) and init <= r['it'] < warmup - term)
        # a window end restarts the step-size search (next start kind 3) except at the very end of the warm-up
        if r['wend']: assert r['next'] == 3 and r['phase'] == 1
        assert r['draw'] == (r['it'] - warmup if r['it'] >= warmup else -1)
    starts = [init] + [e + 1 for e in ends[:-1]]
    for a, e in zip(starts, ends):
        assert [r['wn'] for r in rows if r['it'] == e] == [float(e - a + 1)]
    assert rows[-1]['next'] == 0 and rows[-1]['phase'] == 3      # the chain is done after the last draw
    assert tot['n_post'] == 5 and tot['n_leap_total'] == 7 * (warmup + 5)


def test_dual_averaging_recurrences(tmp_path):
    """stepsize_adaptation::learn_stepsize: x_bar and eps after every warm-up transition, restarted at the window ends with
    mu = log(10 eps); after the last warm-up iteration eps = exp(x_bar) (complete_adaptation)."""
    warmup = 150
    head, rows, tot = _run(tmp_path, warmup)
    delta, gamma, t0, kappa = 0.8, 0.05, 10.0, 0.75
    mu, cnt, sbar, xbar = math.log(10.0), 0, 0.0, 0.0
    for r in rows[:warmup]:
        acc = min(1.0, 0.5 + 0.45 * math.sin(0.37 * r['it']))
        cnt += 1
        eta = 1.0 / (cnt + t0)
        sbar = (1.0 - eta) * sbar + eta * (delta - acc)
        x = mu - sbar * math.sqrt(cnt) / gamma
        x_eta = cnt ** -kappa
        xbar = (1.0 - x_eta) * xbar + x_eta * x
        eps = math.exp(x) if r['it'] + 1 < warmup else math.exp(xbar)
        assert r['eps'] == pytest.approx(eps, rel=1e-13) and r['xbar'] == pytest.approx(xbar, rel=1e-13, abs=1e-15)
        assert r['counter'] == cnt
        if r['wend']:
            mu, cnt, sbar, xbar = math.log(10.0 * eps), 0, 0.0, 0.0
    assert tot['n_div'] == 0 or tot['n_div'] <= 1                 # divergences are counted in sampling only
