#!/usr/bin/env python3
"""Per-wave time stamps of the config-3 passes (needs a -DSL_DEBUG build: VMR_LIB; development aid; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VMR_DEBUG_TIMES", "/tmp/wave_times.txt")
import numpy as np
import torch
from bench import draw_state, CONFIGS
from vimure_amd import CaviEngine
from vimure_amd.synthetic import standard_sbm
cfg = CONFIGS["c3"]
net = standard_sbm(N=cfg["N"], M=cfg["M"], L=cfg["L"], K=cfg["K"], C=2, avg_degree=5.0, sparsify=True, eta=cfg["eta"], seed=0, device="cuda:0")
eng = CaviEngine(net.X, None, K=cfg["K"], mutuality=True, device=0)
sum_x, cov = eng.data_stats()
host, pr = draw_state(cfg, 1, sum_x, cov)
eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
eng.step(6)
eng.sync()
eng.close()
