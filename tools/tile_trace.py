"""Debug (GPU box): per-wave timeline of the S1 tile inside the raw logp kernel (clock64 stamps at phase boundaries)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayes_drt_amd import _lib
from bayes_drt_amd.model import Problem
from bench import synth_spectra, K
from bayes_drt_amd import matrices as gm

lib = _lib.require_gpu()
f, Z = synth_spectra(64)
bf = np.logspace(10, -6, K); tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
prob = Problem([dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)], Z, f, ups_alpha=1.0, ups_beta=0.1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nwg = (B + 15) // 16
th = torch.empty(B, prob.D, dtype=torch.float64, device='cuda').uniform_(-2, 2)
g = torch.empty_like(th); lp = torch.empty(B, dtype=torch.float64, device='cuda')
spec = torch.randint(0, 64, (B,), dtype=torch.int32, device='cuda')
trace = torch.zeros(nwg, 8, 16, dtype=torch.int64, device='cuda')
ts = torch.cuda.Stream()
fn = lib.bdrt_debug_set_tile_trace; fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
for it in range(3):
    if it == 2: assert fn(trace.data_ptr()) == 0
    assert lib.bdrt_logp_grad_dev(prob.handle, th.data_ptr(), spec.data_ptr(), B, 1, lp.data_ptr(), g.data_ptr(), ts.cuda_stream) == 0
    torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.int64)
names = ['entry', 'P1 done', 'B1 passed', 'step0 start', 'step1 start', 'steps done', 'B2 passed', 'lik done', 'B3 passed',
         'gemmBwd done', 'B4 passed', 'end']
t0 = t[:, :, 0].min(axis=1, keepdims=True)
rel = t[:, :, :12] - t0[:, :, None]
# steady-state workgroups only (skip the first wave of workgroups per CU and the last)
sel = slice(nwg // 4, 3 * nwg // 4) if nwg >= 1024 else slice(0, nwg)
print('mean cycles since workgroup entry, per wave (rows) and checkpoint (columns); B=%d' % B)
print(' ' * 6 + ''.join('%13s' % n[:12] for n in names))
for w in range(8):
    print('wave %d' % w + ''.join('%13.0f' % rel[sel, w, k].mean() for k in range(12)))
dur = rel[sel, :, 11].max(axis=1)
print('tile duration: mean %.0f  min %d  max %d cycles' % (dur.mean(), dur.min(), dur.max()))
