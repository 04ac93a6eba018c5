import os, sys, time, warnings, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vimure_amd import CaviEngine, VimureModel
from vimure_amd.synthetic import standard_sbm
warnings.simplefilter("ignore")
L, N, M, K = 4, 2000, 200, 2
net = standard_sbm(N=N, M=M, L=L, K=K, C=2, avg_degree=5.0, sparsify=True, eta=0.5, seed=0, device="cuda:0")
eng = CaviEngine(net.X, None, K=K, mutuality=True, device=0)
for i in range(2):
    m = VimureModel(mutuality=True); t = time.perf_counter(); m.fit(net.X, K=K, seed=1, engine=eng, num_realisations=1, max_iter=500); print("fit %.4f loop %.4f wait %.4f" % (time.perf_counter() - t, m.loop_seconds, m.draw_seconds))
pr = cProfile.Profile(); pr.enable()
m = VimureModel(mutuality=True); m.fit(net.X, K=K, seed=1, engine=eng, num_realisations=1, max_iter=500)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
