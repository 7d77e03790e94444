"""bdrt_logp_grad_dev at B = 1 ... 4096 (81 x 161, inputs in HBM): the one-workgroup-per-point evaluator (BDRT_FEW_POINTS=64: always)
against the 16-column tile evaluator (BDRT_FEW_POINTS=0: always), to place the hand-over between them."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import numpy as np, torch
    from bayes_drt_amd import _lib
    from bayes_drt_amd.model import Problem
    from bench import build_problem_kwargs
    lib = _lib.require_gpu()
    kw = build_problem_kwargs(64)
    blocks, Z, f = kw.pop('blocks'), kw.pop('Z'), kw.pop('freq')
    prob = Problem(blocks, Z, f, **kw)
    ts = torch.cuda.Stream()
    for B in (1, 32, 256, 512, 768, 1024, 1280, 1536, 2048, 4096):
        th = torch.empty(B, prob.D, dtype=torch.float64, device='cuda').uniform_(-2, 2)
        g = torch.empty_like(th); lp = torch.empty(B, dtype=torch.float64, device='cuda')
        spec = torch.randint(0, 64, (B,), dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
        with torch.cuda.stream(ts):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            for it in range(23):
                if it == 3: e0.record(ts)
                assert lib.bdrt_logp_grad_dev(prob.handle, th.data_ptr(), spec.data_ptr(), B, 1, lp.data_ptr(), g.data_ptr(), ts.cuda_stream) == 0
            e1.record(ts)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('%-28s B %5d: %7.2f us per launch, %7.2f M evals/s' % (sys.argv[2], B, us, B / us), flush=True)
else:
    for label, v in (('one workgroup per point', '64'), ('16-column tiles', '0'), ('default dispatch', None)):
        env = dict(os.environ)
        env.pop('BDRT_FEW_POINTS', None)
        if v is not None:
            env['BDRT_FEW_POINTS'] = v
        subprocess.run([sys.executable, __file__, 'child', label], env=env)
