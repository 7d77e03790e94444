"""Randomised cases of the posterior post-processing kernels (bdrt_percentiles, bdrt_summary) against numpy -- the
reference reduces its draws with np.percentile / np.mean (inversion.py:2517-2519, :2560, :2702, :2716-2735, :3068-3113).

Per case (random rows 1 ... 40000, columns 1 ... 400, magnitudes over 40 decades, ties, infinities, NaN columns, random q
including 0 / 100 / repeated values):
  1. column percentiles: bit-equal to np.percentile(X, q, axis=0) (NaN where numpy gives NaN);
  2. projected percentiles: np.percentile(X @ Phi.T + bias, q, axis=0) to 1e-12 of the largest projected value;
  3. summary: mean to 1e-13 relative (a tree sum vs numpy's pairwise sum); percentiles bit-equal for the plain columns and to
     1e-15 relative for the columns that are exponentiated first (the device's exp and glibc's differ in the last bit).

Test infrastructure: `python -m tests.fuzz_post --count 300`."""
import argparse
import sys
import time
import warnings

import numpy as np


def run_case(n):
    from bayes_drt_amd import post
    rng = np.random.default_rng(611953 * n + 29)
    rows = int(rng.choice([1, 2, 3, 7, 40, 400, 800, 4000, 16384, 16385, int(rng.integers(1, 40000))]))
    cols = int(rng.choice([1, 2, 9, 81, 163, int(rng.integers(1, 400))]))
    if rows * cols > 4_000_000:
        cols = max(1, 4_000_000 // rows)
    X = rng.standard_normal((rows, cols)) * np.exp(rng.uniform(-45, 45, cols))
    kind = int(rng.integers(0, 5))
    if kind == 1:
        X[:, 0] = np.round(rng.standard_normal(rows) * 2)                       # ties
    elif kind == 2 and rows > 2:
        X[rng.integers(0, rows), cols - 1] = np.inf; X[rng.integers(0, rows), 0] = -np.inf
    elif kind == 3:
        X[rng.integers(0, rows), rng.integers(0, cols)] = np.nan
    elif kind == 4:
        X[:, cols // 2] = X[0, cols // 2]                                       # a constant column
    nq = int(rng.integers(1, 8))
    q = np.concatenate([rng.uniform(0, 100, nq), rng.choice([0.0, 100.0, 50.0, 2.5, 97.5], 2)])
    rng.shuffle(q)
    text = 'rows=%d cols=%d kind=%d nq=%d' % (rows, cols, kind, len(q))
    fails = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        want = np.percentile(X, q, axis=0)
    got = post.percentile(X, q)
    if got.shape != want.shape or not np.array_equal(got, want, equal_nan=True):
        bad = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
        fails.append('percentile differs at %d entries, first %s: %r vs %r' % (len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])]))
    # projected
    if kind not in (2, 3) and rows <= 8000:
        M = int(rng.integers(1, 220))
        Xp = np.exp(np.clip(rng.standard_normal((rows, cols)), -5, 5))
        Phi = rng.standard_normal((M, cols)); b = rng.standard_normal(M) if rng.random() < 0.5 else None
        Y = Xp @ Phi.T + (0 if b is None else b)
        gp = post.project_percentile(Xp, Phi, b, q)
        wp = np.percentile(Y, q, axis=0)
        if not np.max(np.abs(gp - wp)) <= 1e-12 * np.max(np.abs(Y)):
            fails.append('projected percentile off by %.3g (scale %.3g)' % (np.max(np.abs(gp - wp)), np.max(np.abs(Y))))
    # summary on a log-scale matrix (what the sampler's draws are)
    if kind not in (2, 3):
        T = rng.standard_normal((rows, cols)) * rng.uniform(0.1, 3.0, cols)
        mask = (rng.random(cols) < 0.6).astype(np.uint8)
        mean, pct = post.summary(T, q, mask)
        C_ = np.where(mask[None, :] > 0, np.exp(T), T)
        wm = C_.mean(axis=0)
        if not np.all(np.abs(mean - wm) <= 1e-13 * np.maximum(np.abs(C_).max(axis=0), 1e-300) * max(1, np.log2(rows + 1))):
            fails.append('summary mean off by %.3g relative' % np.max(np.abs(mean - wm) / np.abs(wm)))
        wpct = np.percentile(C_, q, axis=0)
        plain = mask == 0
        if not (np.array_equal(pct[:, plain], wpct[:, plain]) and np.all(np.abs(pct - wpct) <= 1e-15 * np.abs(wpct))):
            fails.append('summary percentiles differ by %.3g relative' % np.max(np.abs(pct - wpct) / np.abs(wpct)))
    return ('FAIL', text + '\n    ' + '\n    '.join(fails)) if fails else ('ok', text)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--first', type=int, default=0)
    ap.add_argument('--count', type=int, default=100)
    a = ap.parse_args()
    t0 = time.time()
    tally = dict(ok=0, FAIL=0)
    for n in range(a.first, a.first + a.count):
        try:
            st, text = run_case(n)
        except Exception as e:
            import traceback
            st, text = 'FAIL', 'exception %s: %s\n%s' % (type(e).__name__, str(e)[:200], traceback.format_exc()[-500:])
        tally[st] += 1
        print('case %4d %-4s %s' % (n, st, text), flush=True)
    print('TOTAL %d ok, %d FAILED in %.0f s' % (tally['ok'], tally['FAIL'], time.time() - t0))
    return 1 if tally['FAIL'] else 0


if __name__ == '__main__':
    sys.exit(main())
