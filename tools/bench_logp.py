"""Micro-benchmark (GPU box): raw batched log-posterior+gradient kernel at B in {16, 256, 2048, 4096, 16384, 65536},
inputs resident in HBM, timed with events on the stream the kernel is launched on."""
import ctypes as C, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayes_drt_amd import _lib
from bayes_drt_amd.model import Problem
from bench import synth_spectra, K, FLOP_PER_EVAL
from bayes_drt_amd import matrices as gm

lib = _lib.require_gpu()
f, Z = synth_spectra(64)
bf = np.logspace(10, -6, K); tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
prob = Problem([dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)], Z, f, ups_alpha=1.0, ups_beta=0.1)
ts = torch.cuda.Stream()           # a real (non-null) stream: the events below are recorded on it too
stream = ts.cuda_stream
for B in (1, 4, 32, 512, 4096, 16384, 65536):
    th = torch.empty(B, prob.D, dtype=torch.float64, device='cuda').uniform_(-2, 2)
    g = torch.empty_like(th); lp = torch.empty(B, dtype=torch.float64, device='cuda')
    spec = torch.randint(0, 64, (B,), dtype=torch.int32, device='cuda')
    def run():
        rc = lib.bdrt_logp_grad_dev(prob.handle, th.data_ptr(), spec.data_ptr(), B, 1, lp.data_ptr(), g.data_ptr(), stream)
        assert rc == 0
    torch.cuda.synchronize()
    with torch.cuda.stream(ts):
        for _ in range(3): run()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record(ts)
        for _ in range(n): run()
        e1.record(ts)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(json.dumps(dict(B=B, ms=ms, evals_per_s=B / ms * 1e3, tflops=B * FLOP_PER_EVAL / ms * 1e-9,
                          frac_mfma=B * FLOP_PER_EVAL / ms * 1e-9 / 78.6)))

# the same evaluator through the HOST-pointer entry point (bdrt_logp_grad: theta and gradient cross PCIe, B x D x 8 bytes each
# way, pageable numpy buffers): the rate a caller without device buffers sees; never the benchmark's `value`
import time
for B in (4096, 65536):
    thh = np.random.default_rng(0).uniform(-2, 2, (B, prob.D))
    sp = np.random.default_rng(1).integers(0, 64, B).astype(np.int32)
    prob.logp_grad(thh, jacobian=True, spec=sp)
    t0 = time.perf_counter()
    for _ in range(5):
        prob.logp_grad(thh, jacobian=True, spec=sp)
    dt = (time.perf_counter() - t0) / 5
    print(json.dumps(dict(B=B, host_pointer_ms=dt * 1e3, evals_per_s_pcie_inclusive=B / dt,
                          bytes_over_pcie=2 * B * prob.D * 8, pcie_GBps=2 * B * prob.D * 8 / dt / 1e9)))
