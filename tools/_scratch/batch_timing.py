import os, sys, time, pickle, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
warnings.simplefilter("ignore")
from vimure_amd import batch
from vimure_amd.batch import _as_data, _fit_unit
from vimure_amd.synthetic import standard_sbm
from vimure_amd.tensor import SparseTensor, layer_of

def timed_units(payload):
    chunk, K, seeds, kw = payload
    out = []
    for u in chunk:
        t0 = time.perf_counter()
        rows, _ = _fit_unit(u[1], u[2], K, seeds, True, None, u[0], None, kw)
        out.append((u[0], time.perf_counter() - t0, sum(r["seconds"] for r in rows)))
    return out

if __name__ == "__main__":
    sizes = [200, 324, 450, 600]
    units = []
    for v, N in enumerate(sizes):
        net = standard_sbm(N=N, M=N, L=4, K=2, avg_degree=3.0, eta=0.3, seed=v, flag_self_reporter=True)
        X, R = SparseTensor.fromarray(net.X), SparseTensor.fromarray(net.R)
        t0 = time.perf_counter()
        Xd, Rd = _as_data(X, R)
        for l in range(4):
            units.append(((v, l), layer_of(Xd, l), layer_of(Rd, l)))
        print("village", N, "slice time", time.perf_counter() - t0, "R nnz/layer", len(units[-1][2].vals), units[-1][2].subs[0].dtype)
    t0 = time.perf_counter(); b = pickle.dumps(units, protocol=5); print("pickle all", time.perf_counter() - t0, len(b) / 1e6, "MB")
    kw = dict(num_realisations=5, max_iter=101)
    # serial, one thread
    res = timed_units((units[:1], 2, [0], kw))   # warm
    t0 = time.perf_counter(); res = timed_units((units, 2, [0, 1, 2], kw)); dt = time.perf_counter() - t0
    print("serial: %.2f s -> %.1f fits/s" % (dt, 48 / dt))
    for tag, tot, fits in res: print(tag, "unit %.3f s, fits %.3f s, create+close %.3f" % (tot, fits, tot - fits))
