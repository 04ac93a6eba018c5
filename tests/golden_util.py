"""Load tests/golden/*.npz (vectors dumped from the real reference by tools/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    # J_*: ingestion vectors, K_*: generator vectors, N_*: an input on which the reference raises -- not model cases
    return [n for n in names if not n.startswith(("J_", "K_", "N_"))]


def load_case(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "X" not in d:  # COO-stored inputs
        shape = tuple(int(s) for s in d["X_shape"])
        X = np.zeros(shape, np.uint8)
        X[tuple(d["X_subs"].astype(np.int64))] = d["X_vals"]
        if "R_subs" in d:
            R = np.zeros(shape, np.uint8)
            R[tuple(d["R_subs"].astype(np.int64))] = 1
        else:   # (every reporter may report on every tie)
            R = np.ones(shape, np.uint8)
        d["X"], d["R"] = X, R
    return d


def case_config(d):
    """(K, mutuality_effective, undirected, seed, prior kwargs, fit kwargs, rho_prior)."""
    und = bool(d["undirected"])
    mut = bool(d["mutuality"]) and not und
    priors = {k[len("prior_"):]: d[k] for k in d if k.startswith("prior_")}
    fitargs = {k[len("fitarg_"):]: d[k].item() for k in d if k.startswith("fitarg_")}
    return int(d["K"]), mut, und, int(d["seed"]), priors, fitargs, d.get("rho_prior")
