#!/usr/bin/env python3
"""cProfile of one unit of the batch driver (development aid, GPU box)."""
import cProfile, pstats, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vimure_amd.batch import _fit_unit
from vimure_amd.synthetic import standard_sbm
warnings.simplefilter("ignore")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 450
net = standard_sbm(N=N, M=N, L=1, K=2, avg_degree=3.0, eta=0.3, seed=1, flag_self_reporter=True)
_fit_unit(net.X, net.R, 2, [0], True, None, 0, None, dict(num_realisations=5, max_iter=101))
pr = cProfile.Profile(); pr.enable()
_fit_unit(net.X, net.R, 2, [1, 2, 3], True, None, 0, None, dict(num_realisations=5, max_iter=101))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
