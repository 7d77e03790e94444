"""Host logic (no GPU): the L-BFGS state machine of bayes_drt_amd/csrc/bdrt_lbfgs.h compiled for the CPU and run
on analytic functions (tests/host/lbfgs_harness.cpp).  The product feeds the same state machine GPU evaluations."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lbfgs_state_machine(tmp_path):
    exe = str(tmp_path / 'lbfgs_harness')
    subprocess.check_call(['g++', '-O2', '-std=c++17', os.path.join(ROOT, 'tests/host/lbfgs_harness.cpp'), '-o', exe])
    out = subprocess.check_output([exe]).decode().strip().splitlines()
    res = {}
    for line in out:
        t = line.split()
        res[t[0]] = dict(iters=int(t[1]), evals=int(t[2]), f=float(t[3]), gnorm=float(t[4]), rc=int(t[5]),
                         x=[float(v) for v in t[6:]])
    assert res['rosen2']['rc'] == 0 and res['rosen2']['f'] < 1e-8 and res['rosen2']['iters'] < 200
    assert all(abs(v - 1) < 1e-3 for v in res['rosen2']['x'])
    assert res['rosen10']['rc'] == 0 and res['rosen10']['f'] < 1e-7
    # condition number 1e8, history 5: iteration cap reached (rc 1) but monotone progress (scipy L-BFGS-B, maxcor=5,
    # reaches f = 1.2 in the same 5000 iterations from f0 = 3.2e8)
    assert res['quad']['rc'] == 1 and res['quad']['iters'] == 5000 and res['quad']['f'] < 10.0
    # -inf outside the domain is handled by the line search
    assert res['barrier']['rc'] == 0 and abs(res['barrier']['x'][0] - 1) < 1e-4
