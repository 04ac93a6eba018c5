"""Batch driver for many small fits: (dataset, layer, seed) units, the shape of the reference's Karnataka
experiment (notebooks/python/experiments/karnataka.py:126-318: per village, per layer separately (L = 1),
10 seeds, `fit(X, R=R, K=2, seed=seed, num_realisations=5, max_iter=101)`, :188-191).

One `CaviEngine` per (dataset, layer) holds the data on the GPU; all seeds of that layer reuse it.
Across GPUs the units are sharded by `vimure_amd.multifit` (one process per GPU, ELBO gather at the end).
"""
import time
import warnings
from typing import Dict, Iterable, Sequence

import numpy as np
import pandas as pd

from .engine import CaviEngine
from .model import VimureModel
from .tensor import to_dense_u8


def fit_layers(X, R=None, K=2, seeds: Iterable[int] = range(10), layer_names: Sequence[str] = None, mutuality=True,
               device=None, keep_posteriors=False, **fit_kwargs) -> pd.DataFrame:
    """Fit every layer of one dataset separately for every seed.  Returns one row per (layer, seed):
    layer, seed, elbo (maxL), iters of the best realisation's last trace row, seconds, nu (= G_exp_nu_f),
    and -- keep_posteriors -- the model object of the best seed per layer in `.attrs["best"]`."""
    Xd = to_dense_u8(X, "X")
    Rd = None if R is None else (to_dense_u8(R, "R") != 0).astype(np.uint8)
    L = Xd.shape[0]
    rows, best = [], {}
    for l in range(L):
        Xl = np.ascontiguousarray(Xd[l:l + 1])
        Rl = None if Rd is None else np.ascontiguousarray(Rd[l:l + 1])
        eng = CaviEngine(Xl, Rl, K=K, mutuality=mutuality, eps=float(fit_kwargs.get("EPS", 1e-12)), device=device)
        try:
            for seed in seeds:
                t0 = time.perf_counter()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    m = VimureModel(mutuality=mutuality)
                    m.fit(Xl, R=Rl, K=K, seed=int(seed), engine=eng, **fit_kwargs)
                dt = time.perf_counter() - t0
                name = layer_names[l] if layer_names is not None else l
                rows.append({"layer": name, "seed": int(seed), "elbo": float(m.maxL),
                             "iters": int(m.trace["iter"].max()) if len(m.trace) else 0,
                             "converged": bool(m.trace["reached_convergence"].any()) if len(m.trace) else False,
                             "seconds": dt, "nu": float(m.G_exp_nu_f)})
                if keep_posteriors and (name not in best or best[name].maxL < m.maxL):
                    best[name] = m
        finally:
            eng.close()
    out = pd.DataFrame(rows)
    if keep_posteriors:
        out.attrs["best"] = best
    return out


def fit_datasets(datasets: Dict[str, tuple], K=2, seeds: Iterable[int] = range(10), dist=None, device=None,
                 **fit_kwargs) -> pd.DataFrame:
    """datasets: name -> (X, R, layer_names or None).  Units (dataset, layer) are sharded over the ranks of
    `dist` (one process per GPU), all seeds of a unit run on the rank that holds its data; the per-fit rows
    are gathered on every rank (RCCL / gloo all_gather of [unit, seed, elbo, iters, seconds, nu])."""
    from .multifit import partition
    seeds = list(seeds)
    units = []
    for name in sorted(datasets):
        X = datasets[name][0]
        for l in range(int(X.shape[0])):
            units.append((name, l, float(X.shape[1]) ** 2 * float(X.shape[3])))
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = partition([u[2] for u in units], world)[rank]
    rows = []
    for ui in mine:
        name, l, _ = units[ui]
        X, R, lnames = (tuple(datasets[name]) + (None,))[:3]
        Xl = to_dense_u8(X, "X")[l:l + 1]
        Rl = None if R is None else to_dense_u8(R, "R")[l:l + 1]
        df = fit_layers(Xl, Rl, K=K, seeds=seeds, device=device, **fit_kwargs)
        for _, r in df.iterrows():
            rows.append([ui, r["seed"], r["elbo"], r["iters"], r["seconds"], r["nu"], float(r["converged"])])
    local = np.asarray(rows, dtype=np.float64).reshape(-1, 7)
    if world > 1:
        import torch
        tdev = "cpu" if dist.get_backend() == "gloo" else (f"cuda:{device}" if device is not None else "cuda")
        full = np.full((len(units) * len(seeds), 7), np.nan)
        full[:len(local)] = local
        t = torch.as_tensor(full, device=tdev)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        allrows = np.concatenate([p.cpu().numpy() for p in parts])
        local = allrows[~np.isnan(allrows[:, 0])]
    out = pd.DataFrame(local, columns=["unit", "seed", "elbo", "iters", "seconds", "nu", "converged"])
    out["dataset"] = [units[int(u)][0] for u in out["unit"]]
    out["layer"] = [units[int(u)][1] for u in out["unit"]]
    return out.sort_values(["dataset", "layer", "seed"]).reset_index(drop=True)
