#!/usr/bin/env python3
"""Per-kernel times of a sweep on self-reporter-mask (Karnataka-shaped) problems: N = M, R[l,i,j,m] = 1 iff m in {i,j}."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimure_amd import CaviEngine  # noqa: E402
from vimure_amd.synthetic import standard_sbm  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [324, 800]
    for N in sizes:
        net = standard_sbm(N=N, M=N, L=1, K=2, avg_degree=3.0, eta=0.3, seed=0, flag_self_reporter=True)
        eng = CaviEngine(net.X, net.R, K=2, mutuality=True)
        g = np.random.RandomState(0)
        eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
        pr = 1 + 0.01 * g.rand(1, N, N, 2); pr /= pr.sum(-1)[..., None]
        eng.set_state(0.1 + 0.1 * g.rand(1, N), 0.1 + 0.1 * g.rand(1, N), 10 + 10 * g.rand(1, 2), 10 + 10 * g.rand(1, 2),
                      0.7, 1.0 + float(net.X.sum()), pr)
        eng.step(5); eng.sync()
        eng.profile(True)
        import time
        t0 = time.perf_counter(); eng.step(100); eng.sync(); dt = time.perf_counter() - t0
        prof = eng.profile_read()
        print(f"N=M={N} format {eng.data_format()} nnzR {int(net.R.sum())}: {dt / 100 * 1e6:.1f} us/sweep (with event profiling); "
              + ", ".join(f"{k} {v['ms'] / max(1, v['launches']) * 1e3:.1f} us x{v['launches']}" for k, v in prof.items() if v["launches"]), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
