"""Independent (dataset, layer, seed) fits over the GPUs of one node -- one process per GPU.

This is how the reference's drivers parallelise (joblib processes over fits:
notebooks/python/experiments/unreliable_reporters.py:364-373; a shell loop over villages:
src/run-karnataka.sh:5-7): units never exchange data while fitting.  The only collective is
the final gather of [unit id, ELBO] pairs (16 B per unit) -- `torch.distributed` all_gather,
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests -- followed by a
point-to-point hand-over of each winner's posteriors to rank 0.
"""
from typing import Callable, Dict, List, Sequence

import numpy as np


def partition(costs: Sequence[float], world: int) -> List[List[int]]:
    """Static longest-first assignment of units to ranks (cost ~ N^2 M per unit); deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += costs[i]
    return [sorted(x) for x in out]


def gather_elbos(local: Dict[int, float], n_units: int, dist=None, device="cpu") -> np.ndarray:
    """All ranks contribute {unit id: ELBO}; every rank gets the full vector (NaN where missing)."""
    import torch
    vec = torch.full((n_units,), float("nan"), dtype=torch.float64, device=device)
    for i, e in local.items():
        vec[i] = e
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vec.cpu().numpy()
    parts = [torch.empty_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, vec)
    stacked = torch.stack(parts).cpu().numpy()
    out = np.full(n_units, np.nan)
    for row in stacked:
        ok = ~np.isnan(row)
        out[ok] = row[ok]
    return out


def send_arrays_to_root(arrays: Dict[str, np.ndarray], owner: int, dist, device="cpu") -> Dict[str, np.ndarray]:
    """Move a dict of float64 arrays from rank `owner` to rank 0 (send/recv; shapes are sent first)."""
    import torch
    rank = dist.get_rank()
    if owner == 0:
        return arrays if rank == 0 else {}
    names = sorted(arrays) if rank == owner else None
    meta = [names, [arrays[n].shape for n in names]] if rank == owner else None
    box = [meta]
    if rank == owner:
        if not hasattr(dist, "send_object_list"):   # (rank 0 would block forever in recv_object_list)
            raise RuntimeError("torch.distributed.send_object_list is missing: torch >= 1.8 is required for the posterior hand-over")
        dist.send_object_list(box, dst=0)
        for n in names:
            dist.send(torch.as_tensor(np.ascontiguousarray(arrays[n], dtype=np.float64), device=device).reshape(-1), dst=0)
        return {}
    if rank == 0:
        dist.recv_object_list(box, src=owner)
        names, shapes = box[0]
        out = {}
        for n, sh in zip(names, shapes):
            t = torch.empty(int(np.prod(sh)) if len(sh) else 1, dtype=torch.float64, device=device)
            dist.recv(t, src=owner)
            out[n] = t.cpu().numpy().reshape(sh)
        return out
    return {}


def fit_many(units: Sequence[dict], fit_fn: Callable[[dict], dict], costs: Sequence[float] = None, dist=None,
             device="cpu", group_key: str = "dataset") -> dict:
    """Fit every unit on the rank that owns it, gather all ELBOs, and bring the posteriors of the best
    unit of every group (e.g. the best seed of each dataset/layer) to rank 0.

    fit_fn(unit) -> {"elbo": float, "posterior": {name: float64 array}}.
    Returns on every rank {"elbo": [n_units] array, "best": {group: unit id}}; rank 0 additionally gets
    "posteriors": {group: {name: array}}.
    """
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    costs = list(costs) if costs is not None else [1.0] * len(units)
    mine = partition(costs, world)[rank]
    results = {i: fit_fn(units[i]) for i in mine}
    elbo = gather_elbos({i: r["elbo"] for i, r in results.items()}, len(units), dist if world > 1 else None, device)
    groups: Dict[object, List[int]] = {}
    for i, u in enumerate(units):
        groups.setdefault(u.get(group_key, 0), []).append(i)
    best = {}
    for gk, idx in groups.items():
        vals = elbo[idx]
        best[gk] = idx[int(np.nanargmax(vals))]   # first strict maximum, as reference model.py:428
    out = {"elbo": elbo, "best": best}
    owners = {}
    parts = partition(costs, world)
    for r_, lst in enumerate(parts):
        for i in lst:
            owners[i] = r_
    post = {}
    for gk in sorted(best, key=str):
        i = best[gk]
        if world == 1:
            post[gk] = results[i]["posterior"]
        else:
            got = send_arrays_to_root(results[i]["posterior"] if owners[i] == rank else {}, owners[i], dist, device)
            if rank == 0:
                post[gk] = got
    if rank == 0:
        out["posteriors"] = post
    return out
