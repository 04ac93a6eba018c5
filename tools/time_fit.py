#!/usr/bin/env python3
"""Where `VimureModel.fit` spends its time on BASELINE config 3 (development aid; run on the GPU box)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vimure_amd import CaviEngine, VimureModel
from vimure_amd.synthetic import standard_sbm
from vimure_amd import _hostlib

L, N, M, K = 4, 2000, 200, 2
net = standard_sbm(N=N, M=M, L=L, K=K, C=2, avg_degree=5.0, sparsify=True, eta=0.5, seed=0, device="cuda:0")
eng = CaviEngine(net.X, None, K=K, mutuality=True, device=0)
sum_x, cov = eng.data_stats()
t = time.perf_counter(); buf = eng.staging(0); print("pinned staging alloc %.3f s" % (time.perf_counter() - t))
p = np.random.RandomState(1)
t = time.perf_counter(); _hostlib.draw_pr_rho(p, (L, N, N, K), 0.0, cov, out=buf); print("C draw into pinned %.3f s" % (time.perf_counter() - t))
p = np.random.RandomState(1)
t = time.perf_counter(); u = p.rand(L, N, N, K); print("numpy rand alone %.3f s" % (time.perf_counter() - t))
for R in (1, 1, 5):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VimureModel(mutuality=True)
        t = time.perf_counter()
        m.fit(net.X, K=K, seed=1, engine=eng, num_realisations=R, max_iter=500)
        dt = time.perf_counter() - t
    print("fit R=%d: %.3f s  (loops %.3f s, waited for states %.3f s), iterations %s" % (R, dt, m.loop_seconds, m.draw_seconds,
          m.trace.groupby("realisation")["iter"].max().tolist()))
