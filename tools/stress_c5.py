#!/usr/bin/env python3
"""BASELINE config 5, one GPU's share: ONE layer of L=8, N=8000, M=1000, K=3 (64 GB of X resident in HBM).
Times sweeps, reports per-kernel rates, and (--parity) checks one sweep + ELBO against the C oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=8000)
    ap.add_argument("--M", type=int, default=1000)
    ap.add_argument("--K", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--parity", action="store_true")
    ap.add_argument("--burn", type=int, default=0, help="sweeps before the timed ones (the first sweeps of this regime leave many all-zero rho rows)")
    ap.add_argument("--zero-rows", action="store_true", help="report the fraction of ties whose rho is all zero after the timed sweeps")
    a = ap.parse_args()
    import torch
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import draw_state
    t0 = time.time()
    net = standard_sbm(N=a.N, M=a.M, L=1, K=a.K, C=2, avg_degree=5.0, eta=0.5, seed=0, device="cuda:0")
    nnz = sum(int(torch.count_nonzero(net.X[0, i:i + 250]).item()) for i in range(0, a.N, 250))
    torch.cuda.empty_cache()
    print(f"generated X {tuple(net.X.shape)} in {time.time() - t0:.1f} s, nnz {nnz} ({100.0 * nnz / net.X.numel():.2f} %)", flush=True)
    t0 = time.time()
    eng = CaviEngine(net.X, None, K=a.K, mutuality=True, device=0)
    print(f"engine created in {time.time() - t0:.1f} s; torch allocated {torch.cuda.memory_allocated() / 1e9:.1f} GB", flush=True)
    sum_x, cov = eng.data_stats()
    cfg = dict(L=1, N=a.N, M=a.M, K=a.K, mutuality=True)
    host, pr = draw_state(cfg, 1, sum_x, cov)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
    try:
        e1 = eng.step(1, want_elbo=True)
    except ValueError:   # (timing experiments with VMR_DEBUG give wrong numbers, possibly NaN)
        e1 = float("nan")
    if a.burn:
        eng.step(a.burn)
    eng.profile(True)
    t0 = time.perf_counter()
    eng.step(a.steps)
    eng.sync()
    dt = time.perf_counter() - t0
    try:
        eng.step(1, want_elbo=True)   # (outside the timed sweeps: the ELBO variant of the rho pass, for the per-kernel table)
    except ValueError:
        pass
    prof = eng.profile_read()
    out = {"shape": [1, a.N, a.N, a.M], "K": a.K, "nnz": nnz, "sweeps_per_s": a.steps / dt, "ms_per_sweep": 1e3 * dt / a.steps,
           "elbo_after_1": e1,
           "kernels": {k: {"avg_ms": v["ms"] / max(1, v["launches"]),
                           "TBps": v["bytes_per_launch"] / (v["ms"] / max(1, v["launches"]) * 1e-3) / 1e12 if v["ms"] > 0 and v["bytes_per_launch"] else None}
                       for k, v in prof.items() if v["launches"]}}
    if a.zero_rows:
        rho = eng.get_state(rho=True)["rho"]
        out["zero_rho_rows"] = float((rho == 0.0).all(axis=-1).mean())
        out["rho_row_sum_min"] = float(rho.sum(axis=-1).min())
        del rho
    if a.parity:
        # the coordinate-list oracle: no dense 64 GB tensor on the host
        from oracle import cavi_coo
        from tests.test_hip_configs import _coo_from_device
        subs, vals = _coo_from_device(net.X)
        del net
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        c = cavi_coo.CooRef((subs, vals), None, (1, a.N, a.N, a.M), a.K, True, (0.1, 0.1, 10.0, 10.0, 0.5, 1.0), host.gamma_shp,
                            host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
        print(f"oracle prepared in {time.perf_counter() - t0:.1f} s", flush=True)
        t0 = time.perf_counter()
        c.cavi_step()
        tc = time.perf_counter() - t0
        e_cpu = c.elbo()
        eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
        e_gpu = eng.step(1, want_elbo=True)
        st = eng.get_state(rho=True)
        out["parity"] = {"cpu_sweep_s": tc, "threads": c.threads(), "elbo_gpu": e_gpu, "elbo_cpu": e_cpu,
                         "elbo_rel_err": abs(e_gpu - e_cpu) / abs(e_cpu),
                         "rho_max_abs": float(np.max(np.abs(st["rho"] - c.rho))),
                         "gamma_shp_rel": float(np.max(np.abs(st["gamma_shp"] - c.gamma_shp) / c.gamma_shp)),
                         "phi_rte_rel": float(np.max(np.abs(st["phi_rte"] - c.phi_rte) / c.phi_rte))}
    print(json.dumps(out), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
