#!/usr/bin/env python3
"""Sweeps per second of the general kernels (csrc/sweep_gen.hip) beside the specialised ones on the same network (development aid;
run on the GPU box): N = 500, M = 50, counts lifted so that K = max(X) + 1 passes 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vimure_amd import CaviEngine
from vimure_amd.synthetic import standard_sbm

net = standard_sbm(N=500, M=50, L=1, K=3, avg_degree=5.0, eta=0.4, seed=1)
X = net.X.astype(np.int64)
g = np.random.RandomState(0)
for K, lift in ((2, 0), (8, 0), (12, 9), (21, 18), (64, 60)):
    Xk = X.copy()
    if lift:
        nz = Xk > 0
        Xk[nz] += g.randint(0, lift, size=int(nz.sum()))
    Xk = np.minimum(Xk, K - 1).astype(np.uint8)
    L, N, _, M = Xk.shape
    pr = 1.0 + 0.01 * g.rand(L, N, N, K)
    pr /= pr.sum(-1)[..., None]
    eng = CaviEngine(Xk, None, K=K, mutuality=True)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    eng.set_state(0.1 + 0.1 * g.rand(L, M), 0.1 + 0.1 * g.rand(L, M), 10 + 10 * g.rand(L, K), 10 + 10 * g.rand(L, K), 0.7, 1.0 + float(Xk.sum()), pr)
    eng.step(3)
    eng.sync()
    t0 = time.perf_counter()
    eng.step(50)
    eng.sync()
    dt = time.perf_counter() - t0
    print(f"K = {K:3d}  reports {int((Xk > 0).sum()):8d}  max count {int(Xk.max()):3d}  {50 / dt:9.1f} sweeps/s  ({1e3 * dt / 50:.3f} ms per sweep)", flush=True)
    eng.close()
