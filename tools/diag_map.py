"""Diagnostic (GPU box): how far apart are two runs of the same L-BFGS fed by GPU vs oracle evaluations?"""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_engine import _bench_problem, _harness, _gamma
from tests.helpers import rel_l2
from bayes_drt_amd.model import Problem
from bayes_drt_amd.engine import optimize_batch
from oracle import oracle as orc

for tag in ('K81', 'K161'):
    blk, Z, f, kw, d = _bench_problem('optimize', tag)
    prob = Problem([blk], Z, f, **kw)
    om = orc.OracleModel([blk], Z, f, **kw)
    th0 = np.random.RandomState(1234).uniform(-2, 2, prob.D)
    h = _harness()
    K = prob.Ks[0]
    for it in (50, 100, 200, 400, 1000, 3000, 10000, 50000):
        t0 = time.time(); out, rep = optimize_batch(prob, th0[None], max_iter=it); tg = time.time() - t0
        ref = np.empty(prob.D); iters = C.c_int(); ne = C.c_int(); lp = C.c_double()
        t0 = time.time()
        h.harness_optimize(C.byref(om.m), th0.ctypes.data_as(C.c_void_p), it, ref.ctypes.data_as(C.c_void_p),
                           C.byref(iters), C.byref(ne), C.byref(lp)); tc = time.time() - t0
        xg, xr = np.exp(out[0][2:2 + K]), np.exp(ref[2:2 + K])
        print(tag, 'max_iter', it, 'gpu iters', rep[0]['iterations'], 'rc', rep[0]['return_code'], 'lp', round(rep[0]['lp'], 4),
              '|g|', '%.2e' % rep[0]['grad_norm'], 'cpu iters', iters.value, 'lp', round(lp.value, 4),
              'gamma rel-L2 %.3e' % rel_l2(_gamma(d, xg), _gamma(d, xr)), 'theta maxdiff %.2e' % np.max(np.abs(out[0] - ref)),
              't_gpu %.2fs t_cpu %.2fs' % (tg, tc), flush=True)
