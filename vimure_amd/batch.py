"""Batch driver for many small fits: (dataset, layer, seed) units, the shape of the reference's Karnataka
experiment (notebooks/python/experiments/karnataka.py:126-318: per village, per layer separately (L = 1),
10 seeds, `fit(X, R=R, K=2, seed=seed, num_realisations=5, max_iter=101)`, :188-191).

A small fit is bound by the latency of its dependent kernel launches (3 per sweep, 500 sweeps per fit), not by the
GPU: one fit keeps a few per cent of an MI355X busy.  Units therefore run CONCURRENTLY: `workers` host threads, each
driving its own `CaviEngine` (one handle = one HIP stream; handles are independent), so the kernels of several fits
overlap on the device -- and, because the threads of one process still meet in the HIP runtime's launch path (8 threads:
1.5x one thread), `processes` worker processes beside each other on the same GPU (4 processes x 2 threads: 3.6x), all threads
pulling units from one shared queue.
One engine per (dataset, layer) holds the data; all seeds of that unit reuse it.
Across GPUs the units are sharded by `vimure_amd.multifit` (one process per GPU, ELBO gather at the end).
`karnataka_tables` / `run_karnataka` produce the four result tables of the reference driver (karnataka.py:200-318).
"""
import os
import threading
import time
import warnings
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, Sequence

import numpy as np
import pandas as pd

from .engine import CaviEngine
from .model import VimureModel
from .tensor import engine_data, is_sparse_like, layer_of, to_dense_u8

DEFAULT_WORKERS = 8


def _fit_unit(Xl, Rl, K, seeds, mutuality, device, name, keep, fit_kwargs):
    """All seeds of one (dataset, layer) unit on one engine.  Returns (rows, models or best model)."""
    rows, models = [], []
    eps = float(fit_kwargs.get("EPS", 1e-12))
    eng = _make_engine(Xl, Rl, K, mutuality, device, eps)
    try:
        for seed in seeds:
            t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = VimureModel(mutuality=mutuality)
                m.fit(Xl, R=Rl, K=K, seed=int(seed), engine=eng, **fit_kwargs)
            dt = time.perf_counter() - t0
            rows.append({"layer": name, "seed": int(seed), "elbo": float(m.maxL),
                         "iters": int(m.trace["iter"].max()) if len(m.trace) else 0,
                         "converged": bool(m.trace["reached_convergence"].any()) if len(m.trace) else False,
                         "seconds": dt, "nu": float(m.G_exp_nu_f)})
            if keep == "all":
                models.append((int(seed), m, dt))
            elif keep == "best" and (not models or models[0][1].maxL < m.maxL):
                models = [(int(seed), m, dt)]
    finally:
        eng.close()
    return rows, models


def _make_engine(Xl, Rl, K, mutuality, device, eps):
    if is_sparse_like(Xl):   # coordinate lists go to the device as they are (vmr_create_coo)
        if Rl is not None and not is_sparse_like(Rl):
            from .tensor import SparseTensor
            Rl = SparseTensor.fromarray(np.asarray(Rl) != 0)
        return CaviEngine.from_coo(Xl.subs, Xl.vals, Xl.shape, R=None if Rl is None else Rl.subs, K=K, mutuality=mutuality,
                                   eps=eps, device=device)
    return CaviEngine(Xl, Rl, K=K, mutuality=mutuality, eps=eps, device=device)


class _LockstepFit:
    """One `VimureModel.fit` taken apart so that the realisations of many fits can advance together: what `fit` does before its
    loop (`__init__`), per realisation before (`begin`) and after (`end`) the CAVI loop, and at the end (`finish`) -- the loop
    itself runs for all fits at once (CaviEngine.fit_loop_batch).  Same draws, same bookkeeping, same results as `fit`."""

    def __init__(self, Xl, Rl, K, seed, mutuality, eng, fit_kwargs, need_rho=True):
        self.need_rho = need_rho   # False: nobody will read rho_f (table rows only): no snapshot of the best realisation, no copy
        kw = dict(fit_kwargs)
        pri = {k: kw.pop(k) for k in ("theta_prior", "lambda_prior", "eta_prior", "rho_prior") if k in kw}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = VimureModel(mutuality=mutuality)
            m._check_fit_params(Xl, pri.get("lambda_prior", (10.0, 10.0)), pri.get("theta_prior", (0.1, 0.1)),
                                pri.get("eta_prior", (0.5, 1.0)), pri.get("rho_prior"), int(seed), R=Rl, K=K, engine=eng, **kw)
        if (eng.L, eng.N, eng.M, eng.K, eng.mutuality) != (m.L, m.N, m.M, m.K, bool(m.mutuality)):
            raise ValueError("engine does not match the shape / K / mutuality of this fit")
        m.close()
        m._engine, m._rho_f = None, None
        m.sumX, coverage = eng.data_stats()
        eng.set_priors(m.alpha_theta, m.beta_theta, m.alpha_lambda, m.beta_lambda, m.alpha_mutuality, m.beta_mutuality)
        m.loop_seconds = m.draw_seconds = 0.0
        self.m, self.eng, self.seed0 = m, eng, int(seed)
        self.states = m._initial_states(eng, coverage)
        self.maxL, self.trace, self.best, self.final_seed = -1e10, [], None, m.seed
        self.r = self.seed_r = None

    def draw(self):
        """Draws the next realisation's initial state on the host (into the engine's staging buffer); False when there is none
        left.  May run while the previous realisation's loop is on the GPU: it touches no device state."""
        t0 = time.perf_counter()
        self._item = item = next(self.states, None)
        if item is not None and item[0] > 0 and self.eng.can_upload_ahead():
            # the upload too leaves the critical path (a device slot of its own, r % 2: the previous use of that slot was consumed
            # by set_state two realisations ago; realisation 0 has nothing to hide behind)
            si = self.m._staging_index(self.eng, item[0])
            pr = item[2]["pr_rho"]
            if isinstance(pr, np.ndarray) and np.shares_memory(pr, self.eng.staging(si)):
                dev = self.eng.upload_ahead(si, item[0] % 2)
                if dev is not None:
                    item[2]["pr_rho"] = dev
        self.m.draw_seconds += time.perf_counter() - t0
        return self._item is not None

    def upload(self):
        """The drawn state to the device (vmr_set_state; synchronises, so the staging buffer is free again afterwards)."""
        self.r, self.seed_r, st, self.final_seed = self._item
        self.eng.set_state(st["gamma_shp"], st["gamma_rte"], st["phi_shp"], st["phi_rte"], st["nu_shp"], st["nu_rte"], st["pr_rho"])

    def end(self, rows, elbo):
        m = self.m
        self.trace.extend((self.r, self.seed_r, it_, e_, rt_, rc_) for it_, e_, rt_, rc_ in rows)
        # the small posteriors cross PCIe only when somebody will look at them: the best realisation so far (`*_f`) and the last
        # one (what `gamma_shp` ... hold after `fit`, as in the reference)
        if self.maxL < elbo or self.r == m.num_realisations - 1:
            m._pull_params(self.eng)
        if self.maxL < elbo:
            self.maxL, self.best = elbo, m._params_copy()
            if m.num_realisations > 1 and self.need_rho:
                self.eng.snapshot()

    def finish(self):
        m = self.m
        if self.best is not None:
            if m.num_realisations > 1 and self.need_rho:
                self.eng.restore()
            m._update_optimal_parameters(self.best, self.eng, lazy_rho=not self.need_rho)
        m._change_seed(self.final_seed)
        m.trace = pd.DataFrame(self.trace, columns=["realisation", "seed", "iter", "elbo", "runtime", "reached_convergence"])
        m.maxL = self.maxL
        return m


def fit_units_lockstep(units, K, seeds, mutuality, device, fit_kwargs, workers=DEFAULT_WORKERS, keep=None, width=64, on_model=None,
                       lanes=None):
    """units: [(tag, Xl, Rl)].  Every (unit, seed) fit as `_fit_unit` would run it, but up to `width` units advance together: one
    engine per unit holds its data, the seeds go one after the other, and realisation r of the current seed runs on all engines
    in lockstep (vmr_fit_loop_batch: one launch per kernel and sweep for all of them).  The host's share -- the RandomState draw
    of each initial state and its upload -- runs on `workers` threads.  Returns {tag: (rows, models)} as `_fit_unit`.
    `seconds` of a row is the wall time of its group's pass over that seed divided by the fits in it.  on_model(tag, seed, model,
    seconds), if given, is called (on the worker threads) for every fitted model as soon as its seed is done -- with rho_f on the
    host -- and the models are not kept.
    lanes (default 3, VMR_BATCH_LANES): groups of `width` units are worked on by that many host threads at once, each with its own
    engines, streams and draw threads -- while one group's lockstep loop keeps the GPU busy, the others' uploads, read-backs and
    table rows (the host's share: about half of a group's wall time) run beside it (3000 Karnataka-shaped fits: 248 -> 274 fits/s)."""
    eps = float(fit_kwargs.get("EPS", 1e-12))
    seeds = [int(s) for s in seeds]
    out = {}
    clock = {"engines": 0.0, "prepare": 0.0, "draw": 0.0, "upload": 0.0, "loops": 0.0, "pull": 0.0, "finish": 0.0, "rows": 0.0}   # (VMR_BATCH_TIMING=1 prints it)
    clock_lock = threading.Lock()
    if lanes is None:
        lanes = int(os.environ.get("VMR_BATCH_LANES", "3"))

    def timed(key, fn):
        t = time.perf_counter()
        r = fn()
        with clock_lock:
            clock[key] += time.perf_counter() - t
        return r
    order = sorted(range(len(units)), key=lambda i: -float(units[i][1].shape[1]))   # similar sizes side by side
    from . import _hostlib
    cap_before, _hostlib.max_threads = _hostlib.max_threads, 1   # many draws side by side: one host thread each
    try:
        def run_group(group):
            # (a pool of its own per group: with one pool for all lanes a lane's uploads and read-backs queue behind the other's draws)
            with ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="vmr-draw") as ex:
                run_group_on(group, ex)

        def run_group_on(group, ex):
            engs = timed("engines", lambda: list(ex.map(lambda u: _make_engine(u[1], u[2], K, mutuality, device, eps), group)))
            try:
                res = {u[0]: ([], []) for u in group}
                for seed in seeds:
                    t0 = time.perf_counter()
                    fits = timed("prepare", lambda: list(ex.map(lambda ue: _LockstepFit(ue[0][1], ue[0][2], K, seed, mutuality, ue[1], fit_kwargs, need_rho=keep is not None or on_model is not None), zip(group, engs))))
                    live = list(range(len(fits)))
                    drawn = [ex.submit(fits[i].draw) for i in live]   # realisation 0
                    while live:
                        live = timed("draw", lambda: [i for i, d in zip(live, drawn) if d.result()])   # (the wait for draws the loop did not hide)
                        if not live:
                            break
                        timed("upload", lambda: list(ex.map(lambda i: fits[i].upload(), live)))
                        drawn = [ex.submit(fits[i].draw) for i in live]   # the next realisation's draws run while this one sweeps
                        m0 = fits[live[0]].m
                        t1 = time.perf_counter()
                        loops = CaviEngine.fit_loop_batch([engs[i] for i in live], m0.max_iter, m0.convergence_tol, m0.decision)
                        dt = time.perf_counter() - t1
                        with clock_lock:
                            clock["loops"] += dt

                        def end(il):
                            fits[il[0]].m.loop_seconds += dt
                            fits[il[0]].end(il[1][0], il[1][1])
                        timed("pull", lambda: list(ex.map(end, zip(live, loops))))
                    models = timed("finish", lambda: list(ex.map(lambda f: f.finish(), fits)))
                    dt = (time.perf_counter() - t0) / max(1, len(fits))
                    t_rows = time.perf_counter()
                    if on_model is not None:
                        list(ex.map(lambda um: on_model(um[0][0], seed, um[1], dt), zip(group, models)))
                    for u, m in zip(group, models):
                        rows, kept = res[u[0]]
                        rows.append({"layer": u[0], "seed": seed, "elbo": float(m.maxL),
                                     "iters": int(m.trace["iter"].max()) if len(m.trace) else 0,
                                     "converged": bool(m.trace["reached_convergence"].any()) if len(m.trace) else False,
                                     "seconds": dt, "nu": float(m.G_exp_nu_f)})
                        if keep == "all":
                            kept.append((seed, m, dt))
                        elif keep == "best" and (not kept or kept[0][1].maxL < m.maxL):
                            kept[:] = [(seed, m, dt)]
                    with clock_lock:
                        clock["rows"] += time.perf_counter() - t_rows
                with clock_lock:
                    out.update(res)
            finally:
                for e in engs:
                    e.close()
        groups = [[units[i] for i in order[g0:g0 + max(1, width)]] for g0 in range(0, len(order), max(1, width))]
        if lanes <= 1 or len(groups) <= 1:
            for group in groups:
                run_group(group)
        else:
            with ThreadPoolExecutor(max_workers=lanes, thread_name_prefix="vmr-lane") as lane_ex:
                for f in [lane_ex.submit(run_group, group) for group in groups]:
                    f.result()
    finally:
        _hostlib.max_threads = cap_before
    if os.environ.get("VMR_BATCH_TIMING"):
        print("fit_units_lockstep seconds:", {k: round(v, 3) for k, v in clock.items()}, flush=True)
    return out


def _as_data(X, R):
    """Coordinate containers stay as they are (when the report lists can hold them); everything else becomes dense uint8."""
    Xd = engine_data(X, "X")
    if is_sparse_like(Xd):
        return Xd, R
    Rd = None if R is None else (to_dense_u8(R, "R") != 0).astype(np.uint8)
    return Xd, Rd


def _run_units(jobs, workers):
    """jobs: callables; run on `workers` threads (1 = in this thread), results in job order."""
    if workers <= 1 or len(jobs) <= 1:
        return [j() for j in jobs]
    with ThreadPoolExecutor(max_workers=min(workers, len(jobs)), thread_name_prefix="vmr-fit") as ex:
        return [f.result() for f in [ex.submit(j) for j in jobs]]


# ---- worker processes (several per GPU) ------------------------------------------------------------------------
# A persistent group of spawned processes (a forked child must not inherit an initialised GPU runtime), each running
# `workers` threads that pull units from ONE shared queue -- so the schedule balances at the grain of a thread -- and
# push (tag, rows) back.  Units travel compactly: coordinate lists as uint16 / int32 columns and uint8 counts (a
# self-reporter mask of N = 600 is 720 k entries: 23 MB as int64 columns, 6 MB on the wire).
class _Coo:
    """What a worker rebuilds from the wire: duck-typed coordinate container (subs, vals, shape)."""

    def __init__(self, subs, vals, shape):
        self.subs, self.vals, self.shape = tuple(subs), vals, tuple(shape)


def _wire(A):
    if A is None or not is_sparse_like(A):
        return A
    shape = tuple(int(v) for v in A.shape)
    subs = tuple(np.ascontiguousarray(a, dtype=np.uint16 if shape[d] <= 65536 else np.int32) for d, a in enumerate(A.subs))
    vals = np.asarray(A.vals)
    if len(vals) == 0 or (vals.min() >= 0 and vals.max() <= 255):
        vals = vals.astype(np.uint8)
    return ("coo", subs, vals, shape)


def _unwire(W):
    if isinstance(W, tuple) and len(W) == 4 and isinstance(W[0], str) and W[0] == "coo":
        return _Coo(W[1], W[2], W[3])
    return W


def _worker_main(tasks, results, workers):
    warnings.simplefilter("ignore")

    def loop():
        while True:
            t = tasks.get()
            if t is None:
                tasks.put(None)   # let the other threads (and processes) see it too
                return
            call, tag, Xw, Rw, K, seeds, mutuality, device, fit_kwargs = t
            try:
                rows, _ = _fit_unit(_unwire(Xw), _unwire(Rw), K, seeds, mutuality, device, tag, None, fit_kwargs)
                results.put((call, tag, rows, None))
            except BaseException as e:   # the caller re-raises
                results.put((call, tag, None, f"{type(e).__name__}: {e}"))
    th = [threading.Thread(target=loop, name=f"vmr-fit-{i}") for i in range(max(1, workers))]
    for t in th:
        t.start()
    for t in th:
        t.join()


class _WorkerGroup:
    def __init__(self, processes, workers):
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        self.tasks, self.results, self.calls = ctx.Queue(), ctx.Queue(), 0
        self.procs = [ctx.Process(target=_worker_main, args=(self.tasks, self.results, workers), daemon=True) for _ in range(processes)]
        for p in self.procs:
            p.start()

    def run(self, units, K, seeds, mutuality, device, fit_kwargs):
        self.calls += 1
        for tag, Xl, Rl in units:
            self.tasks.put((self.calls, tag, _wire(Xl), _wire(Rl), K, seeds, mutuality, device, fit_kwargs))
        out = {}
        while len(out) < len(units):
            try:
                call, tag, rows, err = self.results.get(timeout=5.0)
            except Exception:   # queue.Empty: make sure somebody is still working
                if not all(p.is_alive() for p in self.procs):
                    raise RuntimeError("a vimure_amd worker process died")
                continue
            if call != self.calls:
                continue   # (left over from a call that raised)
            if err is not None:
                raise RuntimeError(f"unit {tag}: {err}")
            out[tag] = rows
        return out

    def shutdown(self):
        self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()


_pools = {}


def _pool(processes, workers):
    if (processes, workers) not in _pools:
        _pools[(processes, workers)] = _WorkerGroup(processes, workers)
    return _pools[(processes, workers)]


def shutdown_pools():
    for p in _pools.values():
        p.shutdown()
    _pools.clear()


def _run_units_processes(units, K, seeds, mutuality, device, processes, workers, fit_kwargs):
    """units: [(tag, Xl, Rl)] sorted longest first, each a task of the worker group.  Returns {tag: rows}."""
    return _pool(processes, max(1, workers)).run(units, K, list(seeds), mutuality, device, fit_kwargs)


def fit_layers(X, R=None, K=2, seeds: Iterable[int] = range(10), layer_names: Sequence[str] = None, mutuality=True,
               device=None, keep_posteriors=False, workers=DEFAULT_WORKERS, **fit_kwargs) -> pd.DataFrame:
    """Fit every layer of one dataset separately for every seed.  Returns one row per (layer, seed):
    layer, seed, elbo (maxL), iters of the best realisation's last trace row, seconds, nu (= G_exp_nu_f),
    and -- keep_posteriors -- the model object of the best seed per layer in `.attrs["best"]`.
    Layers run concurrently on up to `workers` threads (one engine and stream each)."""
    Xd, Rd = _as_data(X, R)
    seeds = list(seeds)
    jobs = []
    for l in range(int(Xd.shape[0])):
        Xl = layer_of(Xd, l)
        Rl = None if Rd is None else layer_of(Rd, l)
        name = layer_names[l] if layer_names is not None else l
        jobs.append(lambda Xl=Xl, Rl=Rl, name=name: _fit_unit(Xl, Rl, K, seeds, mutuality, device, name,
                                                              "best" if keep_posteriors else None, fit_kwargs))
    res = _run_units(jobs, workers)
    out = pd.DataFrame([r for rows, _ in res for r in rows])
    if keep_posteriors:
        out.attrs["best"] = {rows[0]["layer"]: models[0][1] for rows, models in res if models}
    return out


def fit_datasets(datasets: Dict[str, tuple], K=2, seeds: Iterable[int] = range(10), dist=None, device=None,
                 workers=DEFAULT_WORKERS, processes=0, lockstep=True, **fit_kwargs) -> pd.DataFrame:
    """datasets: name -> (X, R, layer_names or None).  Units (dataset, layer) are sharded over the ranks of
    `dist` (one process per GPU), all seeds of a unit run on the rank that holds its data, units of a rank run
    concurrently on `workers` threads -- in `processes` worker processes on the rank's GPU when processes > 0 (keep
    processes x ranks per GPU small: a handful); the per-fit rows are gathered on every rank (RCCL / gloo all_gather of
    [unit, seed, elbo, iters, seconds, nu]).  lockstep (the default without worker processes): the units of a rank advance
    together, every sweep one launch per kernel for all of them (`fit_units_lockstep`); lockstep=False: one engine and stream
    per host thread, as before."""
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
    mutuality = fit_kwargs.pop("mutuality", True)

    def data_of(ui):
        name, l, _ = units[ui]
        X, R, _ = (tuple(datasets[name]) + (None,))[:3]
        if is_sparse_like(X):
            Xd, Rd = _as_data(X, R)
            return layer_of(Xd, l), (None if Rd is None else layer_of(Rd, l))
        return _as_data(np.asarray(X)[l:l + 1], None if R is None else np.asarray(R)[l:l + 1])   # dense: slice first, convert the slice

    def job(ui):
        Xl, Rl = data_of(ui)
        rows, _ = _fit_unit(Xl, Rl, K, seeds, mutuality, device, units[ui][1], None, fit_kwargs)
        return rows
    # longest units first: the tail of the schedule is short fits
    order = sorted(mine, key=lambda ui: -units[ui][2])
    if processes <= 0 and lockstep and len(order) > 1:
        by_tag = fit_units_lockstep([(ui,) + tuple(data_of(ui)) for ui in order], K, seeds, mutuality, device, fit_kwargs, workers=workers)
        res = [(ui, by_tag[ui][0]) for ui in order]
    elif processes > 0:
        by_tag = _run_units_processes([(ui,) + tuple(data_of(ui)) for ui in order], K, seeds, mutuality, device, processes, workers,
                                      fit_kwargs)
        res = [(ui, by_tag[ui]) for ui in order]
    else:
        res = list(zip(order, _run_units([lambda ui=ui: job(ui) for ui in order], workers)))
    local = np.asarray([[ui, r["seed"], r["elbo"], r["iters"], r["seconds"], r["nu"], float(r["converged"])]
                        for ui, rows in res for r in rows], dtype=np.float64).reshape(-1, 7)
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


# ------------------------------------------------------------------------------------------------------------------
# The reference's Karnataka result tables (karnataka.py:200-318), vectorised
# ------------------------------------------------------------------------------------------------------------------
KARNATAKA_FILES = {"summary": "vimure_model_summary.csv", "trace": "vimure_model_trace.csv",
                   "edgelist": "vimure_model_edgelist.csv", "reliability": "vimure_model_reliability.csv"}


def karnataka_tables(model, X, R, village, layer, seed, running_time) -> Dict[str, pd.DataFrame]:
    """The four tables `karnataka.main` appends per (village, layer, seed) -- same columns, same row filter
    (karnataka.py:200-318) -- from a fitted single-layer model and its data X, R ([1,N,N,N] dense uint8)."""
    L, N = int(X.shape[0]), int(X.shape[1])
    if is_sparse_like(X):   # per-tie sums and the ego's / alter's own reports from the coordinate lists
        xl, xi, xj, xm = (np.asarray(a) for a in X.subs)
        xv = np.asarray(X.vals)
        sumX = np.zeros((L, N, N), np.int64)
        np.add.at(sumX, (xl, xi, xj), xv)
        src_rep = np.zeros((L, N, N), bool)
        tgt_rep = np.zeros((L, N, N), bool)
        es, et = (xm == xi) & (xv == 1), (xm == xj) & (xv == 1)
        src_rep[xl[es], xi[es], xj[es]] = True
        tgt_rep[xl[et], xi[et], xj[et]] = True
        reporters = np.unique(np.asarray(R.subs[3]) if is_sparse_like(R) else np.nonzero(np.asarray(R))[3])
    else:
        X = np.asarray(X)
        sumX = X.sum(axis=3)
        ar = np.arange(N)
        src_rep = X[:, ar[:, None], ar[None, :], ar[:, None]] == 1     # X[l,i,j,i]
        tgt_rep = X[:, ar[:, None], ar[None, :], ar[None, :]] == 1     # X[l,i,j,j]
        reporters = np.unique(np.asarray(R.subs[3]) if is_sparse_like(R) else np.nonzero(np.asarray(R))[3])
    summary = pd.DataFrame({"running_time": running_time, "num_realisations": model.num_realisations, "max_iter": model.max_iter,
                            "initial_seed": seed, "best_seed": model.seed, "best_elbo": model.maxL, "eta_est": model.G_exp_nu,
                            "lambda_k": model.G_exp_lambda_f.tolist(), "model": "ViMuRe_T", "village": village, "layer": layer},
                           index=list(range(len(model.G_exp_lambda_f.tolist()))))
    trace = model.trace.copy()
    trace["model"], trace["village"], trace["layer"] = "ViMuRe_T", village, layer
    # edge list: union / intersection baselines and the thresholded posterior (utils.apply_rho_threshold)
    union = sumX > 0
    inter = sumX == 2
    Yv = model.get_inferred_model(method="heuristic_threshold") == 1
    ll, ii, jj = np.nonzero(union | inter | Yv)
    rho1 = model.rho_f[..., 1]
    edgelist = pd.DataFrame({
        "village": village, "layer": layer, "initial_seed": seed, "source": ii, "target": jj,
        "dyad_ID": [f"{i}_{j}" for i, j in zip(ii, jj)],
        "source_report": src_rep[ll, ii, jj], "target_report": tgt_rep[ll, ii, jj],
        "vimure_posterior_probability": rho1[ll, ii, jj],
        "in_union": union[ll, ii, jj], "in_intersection": inter[ll, ii, jj], "in_vimure": Yv[ll, ii, jj],
        "reciprocated_in_union": union[ll, jj, ii], "reciprocated_in_intersection": inter[ll, jj, ii],
        "reciprocated_in_vimure": Yv[ll, jj, ii]})
    lm, mm = np.meshgrid(np.arange(L), np.arange(N), indexing="ij")
    reliability = pd.DataFrame({
        "village": village, "layer": layer, "initial_seed": seed, "node": mm.ravel(),
        "theta": model.G_exp_theta_f[lm.ravel(), mm.ravel()],
        "lambda_theta": model.G_exp_lambda_f[lm.ravel(), 1] * model.G_exp_theta_f[lm.ravel(), mm.ravel()],
        "is_node_reporter": np.isin(mm.ravel(), reporters)})
    return {"summary": summary, "trace": trace, "edgelist": edgelist, "reliability": reliability}


def run_karnataka(villages: Dict[str, tuple], seeds: Iterable[int] = range(1, 11), out_dir: str = None, K=2, mutuality=True,
                  num_realisations=5, max_iter=101, device=None, workers=DEFAULT_WORKERS, lockstep=True, **fit_kwargs) -> Dict[str, pd.DataFrame]:
    """`karnataka.main` over many villages: villages = name -> (X [L,N,N,N], R, layer names); every layer is fitted on its own
    (L = 1) for every seed with `fit(X, R=R, K=2, seed=seed, num_realisations=5, max_iter=101)` (karnataka.py:188-191).
    With `out_dir` the four CSVs are appended as the reference does, and (village, layer, seed) rows already in the summary
    file are skipped (its resume logic, karnataka.py:172-181).  Returns the four tables of this call.  lockstep (default): the
    (village, layer) units advance together, one launch per kernel and sweep for all of them (`fit_units_lockstep`), and each
    model's tables are written as soon as its seed is done; lockstep=False: one unit per host thread."""
    seeds = [int(s) for s in seeds]
    done = set()
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        p = os.path.join(out_dir, KARNATAKA_FILES["summary"])
        if os.path.exists(p) and os.path.getsize(p) > 0:
            prev = pd.read_csv(p)
            done = set(map(tuple, prev[["village", "layer", "initial_seed"]].drop_duplicates().values))
    lock = threading.Lock()
    acc = {k: [] for k in KARNATAKA_FILES}

    def unit(village, l, lname, Xl, Rl):
        todo = [s for s in seeds if (village, lname, s) not in done]
        if not todo:
            return
        _, models = _fit_unit(Xl, Rl, K, todo, mutuality, device, lname, "all",
                              dict(fit_kwargs, num_realisations=num_realisations, max_iter=max_iter))
        for seed, m, dt in models:
            tabs = karnataka_tables(m, Xl, Rl, village, lname, seed, dt)
            with lock:
                for k, df in tabs.items():
                    acc[k].append(df)
                    if out_dir is not None:
                        path = os.path.join(out_dir, KARNATAKA_FILES[k])
                        df.to_csv(path, mode="a", header=not os.path.exists(path) or os.path.getsize(path) == 0, index=False)
    def emit(village, lname, Xl, Rl, seed, m, dt):
        tabs = karnataka_tables(m, Xl, Rl, village, lname, seed, dt)
        with lock:
            for k, df in tabs.items():
                acc[k].append(df)
                if out_dir is not None:
                    path = os.path.join(out_dir, KARNATAKA_FILES[k])
                    df.to_csv(path, mode="a", header=not os.path.exists(path) or os.path.getsize(path) == 0, index=False)
    jobs, by_todo = [], {}
    for village in sorted(villages):
        X, R, lnames = (tuple(villages[village]) + (None,))[:3]
        Xd, Rd = _as_data(X, R)
        for l in range(int(Xd.shape[0])):
            lname = lnames[l] if lnames is not None else l
            Xl, Rl = layer_of(Xd, l), layer_of(Rd, l)
            jobs.append((float(Xd.shape[1]) ** 3, lambda village=village, l=l, lname=lname, Xl=Xl, Rl=Rl: unit(village, l, lname, Xl, Rl)))
            todo = tuple(s for s in seeds if (village, lname, s) not in done)
            if todo:
                by_todo.setdefault(todo, []).append(((village, lname), Xl, Rl))
    if lockstep:
        kw = dict(fit_kwargs, num_realisations=num_realisations, max_iter=max_iter)
        for todo, units in by_todo.items():   # (units that share the seeds still to do advance together)
            data = {u[0]: (u[1], u[2]) for u in units}
            fit_units_lockstep(units, K, list(todo), mutuality, device, kw, workers=workers,
                               on_model=lambda tag, seed, m, dt: emit(tag[0], tag[1], data[tag][0], data[tag][1], seed, m, dt))
    else:
        jobs.sort(key=lambda j: -j[0])
        _run_units([j for _, j in jobs], workers)
    return {k: (pd.concat(v, ignore_index=True) if v else pd.DataFrame()) for k, v in acc.items()}
