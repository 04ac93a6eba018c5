"""Karnataka-shaped small fits: time per sweep, per kernel and per fit on one engine (development aid, GPU box)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
warnings.simplefilter("ignore")
from vimure_amd import CaviEngine, VimureModel
from vimure_amd.synthetic import standard_sbm
from vimure_amd.tensor import SparseTensor, layer_of
for N in (200, 324, 600):
    net = standard_sbm(N=N, M=N, L=1, K=2, avg_degree=3.0, eta=0.3, seed=1, flag_self_reporter=True)
    X, R = SparseTensor.fromarray(net.X), SparseTensor.fromarray(net.R)
    eng = CaviEngine.from_coo(X.subs, X.vals, X.shape, R=R.subs, K=2, mutuality=True)
    for rep in range(3):
        m = VimureModel(mutuality=True)
        t0 = time.perf_counter()
        m.fit(X, R=R, K=2, seed=rep, engine=eng, num_realisations=5, max_iter=101)
        dt = time.perf_counter() - t0
    its = int(m.trace["iter"].max())
    sweeps = int(m.trace.groupby("realisation")["iter"].max().sum()) if "realisation" in m.trace else -1
    print(N, "fit %.4f s loop %.4f s draw %.4f s; trace rows %d, last iter %d, total sweeps %d -> %.1f us/sweep in loop" % (
        dt, m.loop_seconds, m.draw_seconds, len(m.trace), its, sweeps, 1e6 * m.loop_seconds / max(1, sweeps)))
    # raw sweeps
    eng.step(5); eng.sync()
    t0 = time.perf_counter(); eng.step(200); eng.sync(); d = time.perf_counter() - t0
    print("   raw step(200): %.1f us/sweep" % (1e6 * d / 200))
    t0 = time.perf_counter()
    for _ in range(20): eng.step(10, want_elbo=True)
    d = time.perf_counter() - t0
    print("   step(10)+elbo x20: %.1f us/sweep" % (1e6 * d / 200))
    eng.profile(True); eng.step(50); eng.sync(); pr = eng.profile_read()
    print("   kernels:", {k: round(1e3 * v["ms"] / max(1, v["launches"]), 1) for k, v in pr.items() if v["launches"]}, "us")
    eng.close()
