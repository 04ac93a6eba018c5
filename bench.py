#!/usr/bin/env python3
"""Headline benchmark: CAVI iterations/sec of the HIP engine on BASELINE.json's config 3
(synthetic L=4, N=2000, M=200, K=2, mutuality on, R dense), one independent seed-fit per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU; weak scaling: every rank fits the
   same dataset from its own seed, the only collective is the final gather of ELBOs over RCCL)

A "step" is one full CAVI sweep (gamma, phi, rho, nu: reference model.py:623-660) with the ELBO
evaluated at the reference cadence (iteration 1 and every 10th, model.py:1036).  Inputs are
resident in HBM before the timed region (vmr_create has already turned the dense count tensor into
report lists: 4 B per non-zero count).  Prints ONE JSON line (rank 0).

roofline: the dominant kernel's ALGORITHMIC bytes under the data format in use (vmr_kernel_bytes: report
lists = 4 B per non-zero count + 4 B per tie + log-prior read + rho write) over its average
launch time from HIP events on the engine's stream; `traffic` = HBM bytes per launch from rocprofv3
FETCH_SIZE x2 + WRITE_SIZE (profiles/, tools/profile_gpu.sh).  `sweep` restates the whole sweep against
SURVEY 8(d)'s canonical DENSE byte model (12.2 GB per iteration), which this format no longer moves.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "c3": dict(L=4, N=2000, M=200, K=2, eta=0.5, avg_degree=5.0, mutuality=True),
    # configs[1]
    "c2": dict(L=1, N=500, M=50, K=2, eta=0.0, avg_degree=5.0, mutuality=False),
    "tiny": dict(L=2, N=64, M=24, K=2, eta=0.5, avg_degree=4.0, mutuality=True),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def draw_state(cfg, seed, sum_x, coverage):
    """RandomState-exact initial state through the host class (reference model.py:458-605)."""
    from vimure_amd.model import VimureModel
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VimureModel(mutuality=cfg["mutuality"])
    m.L, m.N, m.M, m.K = cfg["L"], cfg["N"], cfg["M"], cfg["K"]
    m.alpha_theta, m.beta_theta = 0.1, 0.1
    m.alpha_lambda, m.beta_lambda = 10.0, 10.0
    m.alpha_mutuality, m.beta_mutuality = 0.5, 1.0
    m.rho_prior = None
    m._change_seed(seed)
    pr = m._draw_pr_rho(coverage, 0.0)
    m._draw_gammas(sum_x)
    return m, pr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (bounded sample)")
    ap.add_argument("--format", default="auto", choices=["auto", "dense", "sparse"],
                    help="engine data layout (VMR_FORMAT): report lists unless X is dense; 'dense' forces the tile path")
    ap.add_argument("--no-converge", action="store_true", help="skip the time-to-ELBO-converge fit")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (fresh child processes under
        # torch.distributed.run; this process never touches the GPU) and hand back their exit code
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the engine")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(dev))

    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm

    if args.format != "auto":
        os.environ["VMR_FORMAT"] = args.format
    cfg = CONFIGS[args.config]
    L, N, M, K = cfg["L"], cfg["N"], cfg["M"], cfg["K"]
    t_gen = time.time()
    net = standard_sbm(N=N, M=M, L=L, K=K, C=2, avg_degree=cfg["avg_degree"], sparsify=True, eta=cfg["eta"],
                       seed=0, device=dev)
    R = torch.ones((L, N, N, M), dtype=torch.uint8, device=dev)  # "multiply reported": dense all-ones mask, streamed
    nnz = sum(int(torch.count_nonzero(net.X[l]).item()) for l in range(L))
    torch.cuda.empty_cache()
    t_gen = time.time() - t_gen
    eng = CaviEngine(net.X, R, K=K, mutuality=cfg["mutuality"], device=local)
    sum_x, cov = eng.data_stats()
    fmt, _ = eng.data_format()
    seed = 1 + rank
    host, pr = draw_state(cfg, seed, sum_x, cov)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)

    def run(first_it, n):
        """n sweeps starting at iteration number first_it, ELBO at the reference cadence."""
        last = None
        it = first_it
        while it < first_it + n:
            if it == 1 or it % 10 == 0:
                last = eng.step(1, want_elbo=True)
                it += 1
            else:
                nxt = min(first_it + n, (it // 10 + 1) * 10)
                eng.step(nxt - it)
                it = nxt
        return last

    run(1, args.warmup)
    eng.sync()
    eng.profile(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    elbo = run(args.warmup + 1, args.steps)
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    if elbo is None:
        elbo = eng.elbo()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the path's only exchange: gather the per-fit ELBOs (RCCL over xGMI)
        mine = torch.tensor([elbo], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        elbos = [float(v.item()) for v in allv]
    else:
        elbos = [elbo]

    if rank == 0:
        total_ms = {k: v["ms"] for k, v in prof.items()}
        dom = max((k for k in total_ms if k != "finalize"), key=lambda k: total_ms[k])
        d = prof[dom]
        avg_ms = d["ms"] / max(1, d["launches"])
        ach = d["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.config + ("" if fmt == "sparse" else "_dense"), {}).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "CAVI iterations/sec (L=4 N=2000 M=200 K=2 synthetic, mutuality on)" if args.config == "c3"
                      else f"CAVI iterations/sec ({args.config})",
            "value": world * args.steps / dt, "unit": "iter/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: StandardSBM-style synthetic L={L} N={N} M={M} K={K}, "
                                   f"eta={cfg['eta']}, R dense all-ones, one seed-fit per GPU" if args.config == "c3"
                                   else args.config,
                       "L": L, "N": N, "M": M, "K": K, "mutuality": cfg["mutuality"], "nnz_X": nnz,
                       "data_format": "report lists (4 B per non-zero count)" if fmt == "sparse" else "dense u8 tiles",
                       "elbo_cadence": "iter 1 and every 10th (fused into the rho pass)", "parallelism": f"fits x{world}"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": d["bytes_per_launch"],
                         "byte_model": ("report lists: 4 B x nnz(X) + 4 B x ties + log-prior read + rho write"
                                        if fmt == "sparse" else "dense: X 1 B/elt + R 1 bit/elt + log-prior read + rho write")},
            "kernels": {k: {"avg_ms": v["ms"] / max(1, v["launches"]), "launches": v["launches"],
                            "GBps": (v["bytes_per_launch"] / (v["ms"] / max(1, v["launches"]) * 1e-3) / 1e9)
                            if v["ms"] > 0 and v["bytes_per_launch"] > 0 else None} for k, v in prof.items()},
            "elbo": elbos, "gen_seconds": t_gen,
        }
        # whole-sweep view: SURVEY 8(d)'s canonical DENSE bytes per iteration (three passes over X,R + ELBO share)
        # against the wall time of a sweep.  The engine makes ONE pass per sweep (sufficient statistics) over report
        # lists, so this "equivalent" rate may exceed the HBM peak: it is a statement about bytes avoided.
        V = float(L) * N * N * M
        srho = 8.0 * L * N * N * K
        b_iter = 3.0 * (V + V / 8.0) + 4.0 * srho
        b_elbo = (V + V / 8.0) + 2.0 * srho
        out["sweep"] = {"canonical_bytes_per_iter": b_iter + b_elbo / 10.0,
                        "canonical_equiv_GBps": (b_iter + b_elbo / 10.0) / (dt / args.steps) / 1e9,
                        "iter_per_s_at_8TBps_canonical": HBM_PEAK_GBS * 1e9 / (b_iter + b_elbo / 10.0)}
        if not args.no_converge:
            out["time_to_converge"] = time_to_converge(cfg, net, eng, seed)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity_full_size"] = cpu_baseline(cfg, net, R, host, pr, args.cpu_seconds, eng)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()   # rank 0 may still be in its (untimed) convergence fit
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N ranks, one per GPU, over RCCL.  Refuses (exit code 2) when the
    node has fewer than N devices; never re-execs a process that has initialised the GPU (device_count() does not)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} but this node has {have} GPU(s); refusing to report a {n}-GPU number", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def time_to_converge(cfg, net, eng, seed):
    """BASELINE's second figure: `VimureModel.fit` on the resident dataset until the reference's stop rule fires
    (|dELBO| < 0.1 on two consecutive checks, checks at iteration 1 and every 10th; model.py:1036-1056), for one
    realisation and for five (the next initial state is drawn on host threads while the GPU sweeps).
    Outside the timed region of `value`."""
    import warnings
    from vimure_amd import VimureModel
    out = {}
    for reals in (1, 5):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = VimureModel(mutuality=cfg["mutuality"])
            t0 = time.perf_counter()
            m.fit(net.X, K=cfg["K"], seed=seed, engine=eng, num_realisations=reals, max_iter=2000)
            wall = time.perf_counter() - t0
        d = {"iterations": m.trace.groupby("realisation")["iter"].max().tolist(),
             "converged": bool(m.trace.groupby("realisation")["reached_convergence"].last().all()),
             "loop_seconds": m.loop_seconds, "fit_seconds": wall, "waited_for_initial_states": m.draw_seconds, "elbo": float(m.maxL)}
        if reals == 1:
            out = dict(d, iterations=d["iterations"][0])
        else:
            out["five_realisations"] = d
    out["note"] = ("fit_seconds = loop_seconds + the RandomState draw of pr_rho (bit-exact with the reference, parallel host "
                   "threads), its upload and ONE read-back of rho at the end (the best realisation is kept on the device)")
    return out


def cpu_baseline(cfg, net, R, host, pr, budget_s, eng):
    """The plain-C oracle (oracle/cavi_ref.c, OpenMP) on the same inputs and initial state, timed on
    this box's host cores for a bounded number of sweeps.  The reference's own NumPy path cannot run
    this configuration at all, so kind = "port"."""
    from oracle import cavi_ref
    X = net.X.cpu().numpy()
    Rh = R.cpu().numpy()
    c = cavi_ref.CRef(X, Rh, cfg["K"], cfg["mutuality"], (0.1, 0.1, 10.0, 10.0, 0.5, 1.0),
                      host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
    cores = c.threads()
    t0 = time.perf_counter()
    n = 0
    while True:
        c.cavi_step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 50 or el / n * (n + 1) > 2 * budget_s:
            break
    base = {"value": n / el, "unit": "iter/s", "cores": cores, "kind": "port",
            "sample": f"{n} full sweeps (gamma,phi,rho,nu; no ELBO) of the same {cfg['L']}x{cfg['N']}x{cfg['N']}x{cfg['M']} "
                      f"workload with oracle/cavi_ref.c (OpenMP, {cores} threads), {el:.1f} s"}
    # full-size parity (outside every timed region): the engine, restarted from the same state, after the same
    # n sweeps, against the C oracle's ELBO and posteriors
    e_cpu = c.elbo()
    eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
    e_gpu = eng.step(n, want_elbo=True)
    st = eng.get_state(rho=True)
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(np.abs(np.asarray(b)), 1e-300)))
    parity = {"sweeps": n, "elbo_gpu": e_gpu, "elbo_cpu_oracle": e_cpu, "elbo_rel_err": abs(e_gpu - e_cpu) / abs(e_cpu),
              "gamma_shp_rel": rel(st["gamma_shp"], c.gamma_shp), "gamma_rte_rel": rel(st["gamma_rte"], c.gamma_rte),
              "phi_shp_rel": rel(st["phi_shp"], c.phi_shp), "phi_rte_rel": rel(st["phi_rte"], c.phi_rte),
              "nu_shp_rel": abs(st["nu_shp"] - c.nu_shp) / abs(c.nu_shp),
              "rho_max_abs": float(np.max(np.abs(st["rho"] - c.rho)))}
    return base, parity


if __name__ == "__main__":
    main()
