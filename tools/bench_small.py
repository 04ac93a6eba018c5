#!/usr/bin/env python3
"""Small-fit regime (BASELINE configs 1, 2, 4): wall time per CAVI iteration and per whole fit."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import load_case  # noqa: E402
from vimure_amd import CaviEngine, VimureModel  # noqa: E402
from vimure_amd.synthetic import standard_sbm  # noqa: E402
import vimure_amd.model as vmm  # noqa: E402


def time_steps(X, R, K, mut, n=200):
    eng = CaviEngine(X, R, K=K, mutuality=mut)
    L, N, _, M = X.shape
    g = np.random.RandomState(0)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    pr = 1 + 0.01 * g.rand(L, N, N, K); pr /= pr.sum(-1)[..., None]
    eng.set_state(0.1 + 0.1 * g.rand(L, M), 0.1 + 0.1 * g.rand(L, M), 10 + 10 * g.rand(L, K), 10 + 10 * g.rand(L, K),
                  0.7, 1.0 + float(X.sum()), pr)
    eng.step(10); eng.sync()
    t0 = time.perf_counter(); eng.step(n); eng.sync(); t1 = time.perf_counter()
    e0 = time.perf_counter()
    for _ in range(20):
        eng.step(1, want_elbo=True)
    e1 = time.perf_counter()
    eng.close()
    return (t1 - t0) / n * 1e6, (e1 - e0) / 20 * 1e6


def main():
    warnings.simplefilter("ignore")
    d = load_case("I_karnataka_vil1_money")
    us, use = time_steps(d["X"], d["R"], 2, True)
    print(f"karnataka vil1 money N=M=324: {us:.1f} us/sweep async, {use:.1f} us/sweep+ELBO sync")
    t0 = time.perf_counter()
    m = VimureModel().fit(d["X"], R=d["R"], K=2, seed=1, num_realisations=5, max_iter=101)
    t1 = time.perf_counter()
    print(f"  whole fit (5 realisations x <=101 iters, like karnataka.py:188-191): {t1 - t0:.2f} s, iters {m.trace['iter'].max()} maxL {m.maxL:.4f}")
    d = load_case("G_config1_sbm")
    us, use = time_steps(d["X"], d["R"], 2, True)
    print(f"config 1 N=100 M=10: {us:.1f} us/sweep, {use:.1f} us/sweep+ELBO")
    t0 = time.perf_counter()
    m = VimureModel().fit(d["X"], R=d["R"], K=2, seed=1)
    print(f"  whole fit: {time.perf_counter() - t0:.3f} s (reference 0.37 s), converged at iter {m.trace['iter'].max()}")
    net = standard_sbm(N=500, M=50, L=1, K=2, avg_degree=5.0, eta=0.0, seed=0)
    us, use = time_steps(net.X, np.ones_like(net.X), 2, False)
    print(f"config 2 N=500 M=50 mutuality off: {us:.1f} us/sweep, {use:.1f} us/sweep+ELBO  (reference: 2.09 s/sweep, 20.8 s/ELBO)")
    net = standard_sbm(N=800, M=800, L=1, K=2, avg_degree=5.0, eta=0.3, seed=0, flag_self_reporter=True)
    us, use = time_steps(net.X, net.R, 2, True, n=50)
    print(f"config-4-like N=M=800 self-reporter mask: {us:.1f} us/sweep, {use:.1f} us/sweep+ELBO")


if __name__ == "__main__":
    main()
