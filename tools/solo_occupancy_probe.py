"""Is the one-chain-per-workgroup kernel (bdrt_solo.h) bound by latency or by issue?  At small problem sizes its LDS footprint
lets the hardware keep 2-4 workgroups per CU resident; the rate with 256 / 512 / 768 / 1024 / 2048 units (BDRT_SOLO=1 forces the
kernel) shows how far co-resident workgroups overlap.  Prints evals/s and us per round for each (Nf, K, units)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['BDRT_SOLO'] = '1'
os.environ['BDRT_VERBOSE'] = '1'
from bayes_drt_amd import _lib, matrices as gm
from bayes_drt_amd.engine import Sampler
from bayes_drt_amd.model import Problem

lib = _lib.require_gpu()
rs = np.random.RandomState(1)
for nf, K in ((41, 41), (41, 61), (81, 81), (81, 121), (81, 161)):
    f = np.logspace(6, -2, nf)
    w = 2 * np.pi * f
    dec = (K - nf) // 2 * (8.0 / (nf - 1))
    bf = np.logspace(6 + dec, -2 - dec, K)
    tau = 1 / (2 * np.pi * bf); eps = 1 / np.mean(np.diff(np.log(tau)))
    z = 1.0 + 1 / (1 + (1j * w * 1e-2) ** 0.8) + 1 / (1 + (1j * w * 1e-4) ** 0.7)
    z = z + 0.003 * rs.normal(size=nf) + 0.003j * rs.normal(size=nf)
    z = z / np.std(np.abs(z)) * np.sqrt(nf / 81)
    A = np.vstack([gm.construct_A(f, 'real', tau=tau, epsilon=eps), gm.construct_A(f, 'imag', tau=tau, epsilon=eps)])
    L = [gm.construct_L(bf, tau=tau, epsilon=eps, order=o) for o in (0, 1, 2)]
    prob = Problem([dict(A=A, L0=L[0], L1=L[1], L2=0.75 * L[2], nonneg=True)], np.concatenate([z.real, z.imag])[None], f,
                   ups_alpha=1.0, ups_beta=0.1)
    for n in (256, 512, 768, 1024, 2048):
        try:
            s = Sampler(prob, n, 1000000, 1, 7)
        except Exception as e:
            print(nf, K, n, 'refused:', e); continue
        if s.kind() != 1:
            print(nf, K, n, 'kernel kind', s.kind(), '(not the one-chain kernel)'); s.close(); break
        s.advance(400); s.sync()
        n0 = s.total_leapfrogs(); s.kernel_time(reset=True)
        t0 = time.perf_counter()
        for _ in range(4):
            s.advance(500)
        s.sync()
        dt = time.perf_counter() - t0
        n1 = s.total_leapfrogs()
        print('Nf %3d K %3d D %3d units %5d: %7.2f M evals/s, %6.2f us per round-of-all-units (%.2f us x units/256)' % (
            nf, K, prob.D, n, (n1 - n0) / dt / 1e6, dt / 2000 * 1e6, dt / 2000 * 1e6 / (n / 256)), flush=True)
        s.close()
