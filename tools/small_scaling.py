#!/usr/bin/env python3
"""Small fits: where does a sweep's time go, and do engines on several host threads overlap?  (development aid, GPU box)
Per N: host enqueue time vs total of step(200) on one engine; then T threads x step(200) on T engines."""
import os, sys, time, threading, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import numpy as np
from vimure_amd import CaviEngine
from vimure_amd.synthetic import standard_sbm
from vimure_amd.tensor import SparseTensor
from bench import draw_state


def engine(N, seed):
    net = standard_sbm(N=N, M=N, L=1, K=2, avg_degree=3.0, eta=0.3, seed=seed, flag_self_reporter=True)
    X, R = SparseTensor.fromarray(net.X), SparseTensor.fromarray(net.R)
    eng = CaviEngine.from_coo(X.subs, X.vals, X.shape, R=R.subs, K=2, mutuality=True)
    sum_x, cov = eng.data_stats()
    host, pr = draw_state(dict(L=1, N=N, M=N, K=2, mutuality=True), seed, sum_x, cov)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
    eng.step(3); eng.sync()
    return eng


def main():
  for N in (200, 600):
      engs = [engine(N, s) for s in range(8)]
      e = engs[0]
      t0 = time.perf_counter(); e.step(200); t1 = time.perf_counter(); e.sync(); t2 = time.perf_counter()
      print(f"N={N}: step(200) enqueue {1e6 * (t1 - t0) / 200:.1f} us/sweep, until done {1e6 * (t2 - t0) / 200:.1f} us/sweep", flush=True)
      e.profile(True); e.step(50); e.sync(); pr = e.profile_read(); e.profile(False)
      print("   kernels us:", {k: round(1e3 * v["ms"] / max(1, v["launches"]), 1) for k, v in pr.items() if v["launches"]}, flush=True)
      for T in (1, 2, 4, 8):
          def run(i):
              engs[i].step(200); engs[i].sync()
          th = [threading.Thread(target=run, args=(i,)) for i in range(T)]
          t0 = time.perf_counter()
          for t in th: t.start()
          for t in th: t.join()
          dt = time.perf_counter() - t0
          print(f"   {T} threads x step(200): {dt * 1e3:.1f} ms -> {T * 200 / dt:.0f} sweeps/s in all", flush=True)
      for x in engs: x.close()


if __name__ == "__main__":
    main()
