import os, sys
sys.path.insert(0, '/root/repo')
os.environ['BDRT_FEW_POINTS'] = '0'
import numpy as np
from tests.test_gpu_model import _log_uniform_problem
from bayes_drt_amd.model import Problem
for nf, K in ((41, 51), (81, 101), (53, 81)):
    blk, Z, f, kw = _log_uniform_problem(nf, K)
    th = np.random.default_rng(K).uniform(-2, 2, (21, 2 * K + 9))
    out = {}
    for name, env in (('gen', {}), ('stream', dict(BDRT_STREAM_A='1'))):
        os.environ.pop('BDRT_STREAM_A', None); os.environ.update(env)
        os.environ['BDRT_VERBOSE'] = '1'
        p = Problem([blk], Z, f, **kw)
        print(name, p.evaluator())
        out[name] = p.logp_grad(th, jacobian=True)
        p.close()
    print(nf, K, np.max(np.abs(out['gen'][1] - out['stream'][1])), np.array_equal(out['gen'][1], out['stream'][1]))
