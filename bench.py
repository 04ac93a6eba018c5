#!/usr/bin/env python3
"""Headline benchmark: CAVI iterations/sec of the HIP engine on BASELINE.json's config 3
(synthetic L=4, N=2000, M=200, K=2, mutuality on, R dense), one independent seed-fit per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU; weak scaling: every rank fits the
   same dataset from its own seed, the only collective is the final gather of ELBOs over RCCL)
  python bench.py --config c5 --gpus 8     BASELINE configs[4]: ONE fit of 8 layers N=8000 M=1000 K=3, one layer per rank
                                           (vimure_amd.sharded: a 3-double RCCL all-reduce per sweep); metric = sweeps/s
  python bench.py --config c4 --gpus N     BASELINE configs[3]: Karnataka-shaped villages x 4 layers x seeds sharded over the
                                           ranks (vimure_amd.batch.fit_datasets(dist=...)); metric = fits/s
  Both refuse (exit code 2) on a node with fewer GPUs than asked for.

The timed block of K steps is repeated REPEATS times inside the run; `value` is the median block (min / max in
`timing_blocks`).  After the timed region, under a time budget, the default run adds two blocks measured on the same GPU:
`c5_layer` (one layer of configs[4] at its stated size: ms per sweep, per-kernel rates) and `small_fits` (the configs[3]
workload of tools/bench_batch.py: fits/s from ONE process).

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


# host threads of the small-fit drivers (RandomState draws of the initial states, uploads, read-backs): env VMR_BENCH_WORKERS
HOST_WORKERS = int(os.environ.get("VMR_BENCH_WORKERS", "8"))


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
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS) + ["c4", "c5"])
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps sweeps; value = the median block")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: no HIP events around the kernels in the timed region (roofline then has no live launch time)")
    ap.add_argument("--no-extra", action="store_true", help="skip the c5_layer / small_fits blocks after the timed region")
    ap.add_argument("--extra-seconds", type=float, default=150.0, help="time budget of the extra blocks")
    ap.add_argument("--c5-nodes", type=int, default=8000, help="--config c5: nodes per layer (the stated configuration is 8000; smaller only to rehearse)")
    ap.add_argument("--villages", type=int, default=75, help="--config c4: villages of the Karnataka table (x 4 layers x --seeds fits); 75 = the stated workload")
    ap.add_argument("--seeds", type=int, default=10, help="--config c4: seeds per (village, layer); the reference's driver runs 10")
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
    # VMR_BENCH_REHEARSE=1 (development: the N-rank code path on a box with fewer GPUs): ranks share the devices there are and
    # talk over gloo with host tensors; the line then says so and is no N-GPU measurement
    rehearse = bool(os.environ.get("VMR_BENCH_REHEARSE")) and world > 1
    if rehearse:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    cdev = "cpu" if rehearse else dev   # where the collectives' tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    if world > 1:
        # every rank says where it runs (stderr): a rank that shares a device with another, or sees the wrong one, shows here
        prop = torch.cuda.get_device_properties(local)
        print(f"bench.py: rank {rank}/{world} on cuda:{local} ({prop.name}, {prop.multi_processor_count} CUs, {prop.total_memory >> 30} GiB), "
              f"backend {dist.get_backend()}, world size observed {dist.get_world_size()}", file=sys.stderr, flush=True)
        ids = [None] * world
        dist.all_gather_object(ids, (local, torch.cuda.get_device_properties(local).uuid.__str__() if hasattr(prop, "uuid") else str(local)))
        if rank == 0 and not rehearse and len({u for _, u in ids}) != world:
            raise SystemExit(f"bench.py: {world} ranks but only {len({u for _, u in ids})} distinct devices: {ids}")
    if world > 1 and rank == 0:
        print(f"bench.py: RCCL world size {dist.get_world_size()} (backend {dist.get_backend()}), one rank per GPU", file=sys.stderr, flush=True)
    if args.config == "c5":
        return bench_c5_sharded(args, rank, world, local, dev, dist, N=args.c5_nodes, cdev=cdev)
    if args.config == "c4":
        return bench_c4_batch(args, rank, world, local, dev, dist, cdev=cdev)

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
    if not args.no_kernel_events:
        eng.profile(2)   # events around the passes over the data only (the roofline kernels), not the finalize kernels
    blocks = []
    elbo = None
    it0 = args.warmup + 1
    for _ in range(max(1, args.repeats)):   # every block: barrier + synchronize, EXACTLY --steps sweeps, synchronize + barrier
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e_ = run(it0, args.steps)
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        blocks.append(time.perf_counter() - t0)
        elbo = e_ if e_ is not None else elbo
        it0 += args.steps
    own_ms = 1e3 * float(np.median(blocks)) / args.steps   # this rank's own pace, before the MAX over ranks
    per_rank_ms = [own_ms]
    if dist is not None:   # MAX over ranks, block by block
        tt = torch.tensor(blocks, dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        blocks = [float(v) for v in tt.tolist()]
        pr_ = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(pr_, torch.tensor([own_ms], dtype=torch.float64, device=cdev))
        per_rank_ms = [float(v.item()) for v in pr_]
        if rank == 0 and max(per_rank_ms) > 1.10 * min(per_rank_ms):
            # independent fits of the same dataset (weak scaling): every rank should run at the single-GPU pace
            print(f"bench.py: WARNING ranks differ by more than 10 % in ms per sweep: {['%.4f' % v for v in per_rank_ms]}", file=sys.stderr, flush=True)
    dt = float(np.median(blocks))
    prof = eng.profile_read()
    eng.profile(False)
    if elbo is None:
        elbo = eng.elbo()
    if dist is not None:
        # the path's only exchange: gather the per-fit ELBOs (RCCL over xGMI)
        mine = torch.tensor([elbo], dtype=torch.float64, device=cdev)
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
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: ranks share a GPU, gloo)",
            "config": {"workload": f"BASELINE configs[2]: StandardSBM-style synthetic L={L} N={N} M={M} K={K}, "
                                   f"eta={cfg['eta']}, R dense all-ones, one seed-fit per GPU" if args.config == "c3"
                                   else args.config,
                       "L": L, "N": N, "M": M, "K": K, "mutuality": cfg["mutuality"], "nnz_X": nnz,
                       "data_format": "report lists (4 B per non-zero count)" if fmt == "sparse" else "dense u8 tiles",
                       "elbo_cadence": "iter 1 and every 10th (fused into the rho pass)", "parallelism": f"fits x{world}"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of this command, "
                                           "collected in separate --pmc passes; not re-measured in this run)",
                         "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": d["bytes_per_launch"],
                         "byte_model": ("report lists: 4 B x nnz(X) + 4 B per 64 ties + log-prior read; rho is used (statistics, nu) and NOT "
                                        "written: the next sweep overwrites it unread (the last sweep of a vmr_step call and ELBO sweeps write it)"
                                        if dom == "rho_nostore" else
                                        "report lists: 4 B x nnz(X) + 4 B per 64 ties + log-prior read + rho write"
                                        if fmt == "sparse" else "dense: X 1 B/elt + R 1 bit/elt + log-prior read + rho write")},
            "kernels": {k: {"avg_ms": v["ms"] / max(1, v["launches"]), "launches": v["launches"],
                            "GBps": (v["bytes_per_launch"] / (v["ms"] / max(1, v["launches"]) * 1e-3) / 1e9)
                            if v["ms"] > 0 and v["bytes_per_launch"] > 0 else None} for k, v in prof.items()},
            "elbo": elbos, "gen_seconds": t_gen,
            "timing_blocks": {"repeats": len(blocks), "steps_per_block": args.steps, "median_s": dt, "min_s": min(blocks),
                              "max_s": max(blocks), "value_from": "median block"},
            "per_rank_ms_per_step": per_rank_ms,   # each rank's own pace (weak scaling: every entry should equal the 1-GPU ms_per_step)
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
            try:
                out["cpu_baseline_sparse"] = cpu_baseline_sparse(cfg, net, host, pr, args.cpu_seconds, out["parity_full_size"])
            except Exception as e:
                out["cpu_baseline_sparse"] = {"error": repr(e)}
        eng.close()
        del net, R
        torch.cuda.empty_cache()
        if not args.no_extra and world == 1 and args.config == "c3":
            t_extra = time.perf_counter()
            try:
                out["c5_layer"] = c5_layer_block(local)
            except Exception as e:   # (a smaller GPU, or memory held by someone else: the headline line must still come out)
                out["c5_layer"] = {"error": repr(e)}
            left = args.extra_seconds - (time.perf_counter() - t_extra)
            try:
                out["small_fits"] = small_fits_block(local, left) if left > 10 else {"skipped": "time budget"}
            except Exception as e:
                out["small_fits"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    else:
        eng.close()
    if dist is not None:
        dist.barrier()   # rank 0 may still be in its (untimed) convergence fit
        dist.destroy_process_group()


def c5_layer_block(device, N=8000, M=1000, K=3, sweeps=10):
    """BASELINE configs[4], one GPU's share: ONE layer of L=8, N=8000, M=1000, K=3 (64 GB of X generated on the device).
    Outside the timed region of `value`.  Per kernel class: average launch time (HIP events on the engine's stream),
    algorithmic bytes, fraction of the 8 TB/s HBM peak; `single_pass_model_frac` restates the whole sweep against the bytes ONE
    pass over the lists would move (entries + log prior + rho write)."""
    import torch
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    t0 = time.perf_counter()
    net = standard_sbm(N=N, M=M, L=1, K=K, C=2, avg_degree=5.0, eta=0.5, seed=0, device=f"cuda:{device}")
    eng = CaviEngine(net.X, None, K=K, mutuality=True, device=device)
    del net
    torch.cuda.empty_cache()
    sum_x, cov = eng.data_stats()
    cfg = dict(L=1, N=N, M=M, K=K, mutuality=True)
    host, pr = draw_state(cfg, 1, sum_x, cov)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
    del pr
    t_setup = time.perf_counter() - t0
    eng.step(1, want_elbo=True)
    eng.profile(True)
    t0 = time.perf_counter()
    eng.step(sweeps)
    eng.sync()
    dt = time.perf_counter() - t0
    e = eng.step(1, want_elbo=True)   # (outside the timed sweeps: the ELBO variant, for the per-kernel table)
    prof = eng.profile_read()
    _, nnz = eng.data_format()
    eng.close()
    one_pass = 4.0 * nnz + 2.0 * 8.0 * N * N * K
    kern = {}
    for k, v in prof.items():
        if not v["launches"]:
            continue
        ms = v["ms"] / v["launches"]
        kern[k] = {"avg_ms": ms, "launches": v["launches"], "bytes": v["bytes_per_launch"],
                   "frac_of_8TBps": (v["bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 and v["bytes_per_launch"] else None}
    return {"workload": f"one layer of BASELINE configs[4]: N={N} M={M} K={K}, mutuality on, all-ones mask, generated on the device",
            "nnz": nnz, "ms_per_sweep": 1e3 * dt / sweeps, "sweeps_per_s": sweeps / dt, "kernels": kern,
            "single_pass_bytes": one_pass, "single_pass_model_frac": one_pass / (dt / sweeps) / 1e9 / HBM_PEAK_GBS,
            "elbo_after_sweeps": e, "setup_seconds": t_setup}


def small_fits_block(device, budget_s, sizes=None, n_seeds=None):
    """BASELINE configs[3] shape (tools/bench_batch.py): Karnataka-like villages (self-reporter mask, M-dim = N, 4 layers fitted
    separately, K=2, 5 realisations x <= 101 iterations per fit; karnataka.py:170-191) through vimure_amd.batch from ONE
    process: the (village, layer) units -- 48 here; the reference's experiment has 300 -- advance in lockstep, one launch per kernel and sweep for all of them
    (vmr_fit_loop_batch), the seeds one after the other.  Outside the timed region of `value`."""
    import warnings
    from vimure_amd.batch import fit_datasets
    from vimure_amd.synthetic import standard_sbm
    from vimure_amd.tensor import SparseTensor
    warnings.simplefilter("ignore")
    # the stated workload -- 75 villages x 4 layers x 10 seeds = 3000 fits -- when the time budget allows (about 15 s of fits and as much
    # set-up), else the first 12 villages x 3 seeds
    full = sizes is None and budget_s >= 60.0
    table = KARNATAKA_VILLAGES if full else (KARNATAKA_VILLAGES[:12] if sizes is None else [(n, None) for n in sizes])
    n_seeds = n_seeds if n_seeds is not None else (10 if full else 3)
    sizes = [n for n, _ in table]
    data = {f"vil{v}": village_coo(N, v, f"cuda:{device}", n_resp=nr) for v, (N, nr) in enumerate(table)}
    fit_datasets({"w": data["vil0"]}, K=2, seeds=range(1), num_realisations=1, max_iter=11, workers=4, device=device)   # warm-up
    t0 = time.perf_counter()
    df = fit_datasets(data, K=2, seeds=range(n_seeds), num_realisations=5, max_iter=101, workers=HOST_WORKERS, device=device)
    dt = time.perf_counter() - t0
    sweeps = float(df["iters"].sum())   # (iterations of the best realisation only: a lower bound on the sweeps run)
    return {"workload": f"{len(sizes)} villages (Karnataka table: N = {min(sizes)}..{max(sizes)}, respondents as measured) x 4 layers x {n_seeds} seeds, "
                        "5 realisations x <= 101 iterations each" + (" = BASELINE configs[3] as stated" if full else ""),
            "fits": int(len(df)), "seconds": dt, "fits_per_s": len(df) / dt, "processes": 1, "host_threads": HOST_WORKERS, "lockstep_units": 4 * len(sizes),
            "mean_fit_seconds": float(df["seconds"].mean()), "sweeps_per_s_lower_bound": sweeps / dt}


def bench_c5_sharded(args, rank, world, local, dev, dist, N=8000, M=1000, K=3, cdev=None):
    """BASELINE configs[4]: ONE fit whose layers live on different GPUs, one layer per rank (vimure_amd.sharded)."""
    import torch
    from vimure_amd.sharded import fit_layer_sharded
    from vimure_amd.synthetic import standard_sbm
    if world < 2:
        raise SystemExit("--config c5 shards the LAYERS of one fit over the ranks: --gpus >= 2 (the stated configuration is 8)")
    net = standard_sbm(N=N, M=M, L=1, K=K, C=2, avg_degree=5.0, eta=0.5, seed=rank, device=dev)
    n_sweeps = args.warmup + args.steps
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fit_layer_sharded(net.X, None, [rank], world, K, dist, seed=1, num_realisations=1, max_iter=n_sweeps, device=local)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=cdev or dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        its = res["trace"][-1][2] if res["trace"] else n_sweeps
        print(json.dumps({"metric": f"CAVI sweeps/sec of ONE layer-sharded fit (L={world} N={N} M={M} K={K}, one layer per GPU)",
                          "value": its / float(tt.item()), "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * float(tt.item()) / max(1, its), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": f"BASELINE configs[4]: L={world} N={N} M={M} K={K}, layer-sharded, whole fit incl. set-up of the "
                                                 f"initial state; a 3-double RCCL all-reduce per sweep", "iterations": its, "rccl_world": world},
                          "elbo": res["maxL"], "note": "includes the host draw and upload of the initial state (1.5 GB per layer)"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# The 75 villages of the reference's Karnataka data (data/input/india_microfinance/formatted/vil*_edges.csv), measured
# in the build container: (nodes N = distinct i / j of the edge list, respondents = distinct `respondent`).  Data statistics only --
# the workload shape of BASELINE configs[3]: every village x 4 layers (karnataka.py:15 LAYERS) x 10 seeds (karnataka.py:21) fits.
KARNATAKA_VILLAGES = [
    (198, 94), (326, 142), (390, 195), (337, 150), (472, 212), (369, 178), (448, 199), (661, 284), (559, 243), (417, 203), (343, 159),
    (495, 210), (639, 280), (437, 211), (688, 303), (317, 149), (353, 174), (774, 395), (623, 303), (414, 203), (359, 170), (386, 200),
    (627, 301), (429, 219), (321, 181), (386, 216), (590, 293), (263, 132), (376, 182), (720, 370), (671, 344), (527, 266), (353, 181),
    (415, 206), (447, 227), (519, 258), (541, 263), (577, 279), (364, 160), (447, 217), (387, 184), (506, 255), (522, 261), (620, 309),
    (856, 395), (375, 170), (261, 124), (575, 279), (294, 148), (467, 239), (474, 216), (811, 387), (338, 164), (898, 413), (311, 155),
    (493, 242), (407, 190), (619, 294), (750, 341), (425, 189), (512, 230), (360, 164), (490, 220), (219, 110), (528, 233), (695, 298),
    (525, 238), (447, 217), (386, 193), (472, 210), (579, 269), (338, 172), (388, 172), (231, 109), (523, 246)]


def village_coo(N, seed, dev, L=4, n_resp=None):
    """A Karnataka-shaped village as the reader delivers it -- coordinate containers of X and of the self-reporter mask R (R[l,i,j,m]
    = 1 iff m is i or j AND m is a respondent, `_io.py:230-242`; the reporter dimension is N, `_io.py:230,253`) -- drawn on the
    device (the host generator needs minutes for N = 800) and masked there.  n_resp: how many of the N nodes answered the survey
    (None: all of them)."""
    import torch
    from vimure_amd.synthetic import standard_sbm
    from vimure_amd.tensor import SparseTensor
    net = standard_sbm(N=N, M=N, L=L, K=2, avg_degree=3.0, eta=0.3, seed=seed, device=dev)
    resp = np.ones(N, bool)
    if n_resp is not None and n_resp < N:
        resp[:] = False
        resp[np.random.RandomState(1000 + seed).choice(N, size=int(n_resp), replace=False)] = True
    resp_d = torch.as_tensor(resp, device=dev)
    parts = []
    for l in range(L):   # (layer by layer: torch.nonzero fails beyond 2^31 elements, and 4 x 898^3 is more)
        il = torch.nonzero(net.X[l])   # row-major order, as np.nonzero
        il = il[((il[:, 2] == il[:, 0]) | (il[:, 2] == il[:, 1])) & resp_d[il[:, 2]]]
        parts.append(torch.cat([torch.full((il.shape[0], 1), l, dtype=il.dtype, device=il.device), il], dim=1))
    idx = torch.cat(parts, dim=0)
    vals = net.X[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]].cpu().numpy().astype(np.int64)
    X = SparseTensor(tuple(idx[:, d].cpu().numpy() for d in range(4)), vals, shape=(L, N, N, N))
    del net, idx
    torch.cuda.empty_cache()
    ll, ii, jj = (a.ravel() for a in np.meshgrid(np.arange(L), np.arange(N), np.arange(N), indexing="ij"))
    lo, hi = np.minimum(ii, jj), np.maximum(ii, jj)
    mm = np.stack([lo, hi], axis=1).ravel()
    keep = resp[mm]
    keep[1::2] &= (lo != hi)   # (i == j: one reporter)
    rep2 = lambda a: np.repeat(a, 2)[keep]
    mm = mm[keep]
    R = SparseTensor((rep2(ll), rep2(ii), rep2(jj), mm), np.ones(len(mm), np.int64), shape=(L, N, N, N))
    return X, R


def bench_c4_batch(args, rank, world, local, dev, dist, cdev=None):
    """BASELINE configs[3]: (village, layer, seed) fits sharded over the ranks, gathered with one all_gather."""
    import torch
    import warnings
    from vimure_amd.batch import fit_datasets
    from vimure_amd.synthetic import standard_sbm
    from vimure_amd.tensor import SparseTensor
    warnings.simplefilter("ignore")
    table = KARNATAKA_VILLAGES[:max(1, min(args.villages, len(KARNATAKA_VILLAGES)))]   # the stated workload: all 75
    data = {}
    for v, (N, nresp) in enumerate(table):
        data[f"vil{v:02d}"] = village_coo(int(N), v, dev, n_resp=int(nresp))
    fit_datasets({"w": data["vil00"]}, K=2, seeds=range(1), num_realisations=1, max_iter=11, workers=4, device=local)   # warm-up
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    df = fit_datasets(data, K=2, seeds=range(args.seeds), num_realisations=5, max_iter=101, workers=HOST_WORKERS, device=local,
                      dist=dist if world > 1 else None)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev or dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": "Karnataka-shaped fits/sec (village x layer x seed units, 5 realisations x <= 101 iterations each)",
                          "value": len(df) / dt, "unit": "fits/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / max(1, len(df)), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": "f64", "data": "synthetic",
                          "config": {"workload": f"BASELINE configs[3]: {len(table)} villages (node and respondent counts of the reference's 75 "
                                                 f"Karnataka villages, N = 198..898, reporter dimension N, self-reporter mask) x 4 "
                                                 f"layers x {args.seeds} seeds = {len(df)} fits, sharded over {world} rank(s)",
                                     "rccl_world": world}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N ranks, one per GPU, over RCCL.  Refuses (exit code 2) when the
    node has fewer than N devices; never re-execs a process that has initialised the GPU (device_count() does not)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("VMR_BENCH_REHEARSE"):
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
    t0 = time.perf_counter()
    eng.can_upload_ahead()   # the engine's page-locked staging buffer (256 MB at config 3): allocated once per engine, not per fit
    t_alloc = time.perf_counter() - t0
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
    out["staging_alloc_seconds"] = t_alloc
    out["note"] = ("fit_seconds = loop_seconds + the RandomState draw of pr_rho (bit-exact with the reference, parallel host "
                   "threads), its upload and ONE read-back of rho at the end (the best realisation is kept on the device); the "
                   "engine's page-locked staging buffer is allocated before the timing (staging_alloc_seconds, once per engine: the "
                   "first fit on a fresh engine pays it on top)")
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


def cpu_baseline_sparse(cfg, net, host, pr, budget_s, parity):
    """The like-for-like CPU figure: oracle/cavi_coo.c -- the same algorithm class as the engine (coordinate lists of the non-zero
    counts, all-ones mask implicit; OpenMP over ties) -- on the same inputs and initial state, same box, bounded sample.  Its
    set-up (sorting the lists, the mirror counts) is outside the timing, as vmr_create's is outside the engine's."""
    import torch
    from oracle import cavi_coo
    L, N, M, K = cfg["L"], cfg["N"], cfg["M"], cfg["K"]
    parts, vparts = [], []
    for l in range(L):   # (layer by layer: torch.nonzero fails beyond 2^31 elements)
        il = torch.nonzero(net.X[l])
        vparts.append(net.X[l][il[:, 0], il[:, 1], il[:, 2]].cpu().numpy().astype(np.int32))
        parts.append(np.concatenate([np.full((il.shape[0], 1), l, np.int64), il.cpu().numpy()], axis=1))
        del il
    idx = np.concatenate(parts, axis=0)
    vals = np.concatenate(vparts)
    subs = tuple(np.ascontiguousarray(idx[:, d]) for d in range(4))
    del idx, parts, vparts
    t0 = time.perf_counter()
    c = cavi_coo.CooRef((subs, vals), None, (L, N, N, M), K, cfg["mutuality"], (0.1, 0.1, 10.0, 10.0, 0.5, 1.0),
                        host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
    t_prep = time.perf_counter() - t0
    cores = c.threads()
    t0 = time.perf_counter()
    n = 0
    while True:
        c.cavi_step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200 or el / n * (n + 1) > 2 * budget_s:
            break
    out = {"value": n / el, "unit": "iter/s", "cores": cores, "kind": "port",
           "sample": f"{n} full sweeps (gamma,phi,rho,nu; no ELBO) of the same workload with oracle/cavi_coo.c (coordinate lists of the "
                     f"{len(vals)} non-zero counts, OpenMP, {cores} threads), {el:.1f} s; list set-up {t_prep:.1f} s not counted"}
    if n == parity.get("sweeps"):   # the two oracles after the same number of sweeps from the same state
        out["elbo"] = c.elbo()
        out["elbo_rel_diff_to_dense_oracle"] = abs(out["elbo"] - parity["elbo_cpu_oracle"]) / abs(parity["elbo_cpu_oracle"])
    return out


if __name__ == "__main__":
    main()
