"""One fit whose LAYERS live on different GPUs (BASELINE config 5: L=8, N=8000, M=1000 -- 64 GB of X per layer).

Layers are coupled only through the mutuality scalar nu (reference model.py:589-596, 822-825) and the joint
ELBO stop rule (model.py:1039-1047), so every rank sweeps its own layers with `vmr_sweep_local` and the ranks
all-reduce THREE doubles per sweep (nu partial, and on ELBO iterations two ELBO partials) -- latency-bound,
RCCL over xGMI on GPUs ("nccl"), gloo in the tests.  The result equals a single-handle fit of all layers.

On RCCL the three doubles never leave the device: `vmr_sweep_local_dev` leaves them in a CUDA tensor, the all-reduce is
queued on the engine's own stream (torch.cuda.ExternalStream over `vmr_stream`) and `vmr_commit_nu_dev` reads the total
from device memory -- sweeps are queued back to back and the host only waits at ELBO evaluations (every 10th sweep).
Every rank draws only its OWN layers of the initial state: the RandomState stream of the reference is skipped over the
other layers (`_hostlib.draw_pr_rho_layers`), not drawn and dropped.
"""
import time

import numpy as np
import scipy.special as sp

from . import _hostlib
from .engine import CaviEngine

INF = 1e10


def _allreduce(vals, dist, device):
    import torch
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    dist.all_reduce(t)
    return t.cpu().tolist()


def _gamma_term(pa, pb, qa, qb):   # model.py:1300-1303
    return sp.gammaln(qa) - pa * np.log(qb) + (pa - qa) * sp.psi(qa) + qa * (1.0 - pb / qb)


def fit_layer_sharded(X_local, R_local, local_layers, L_total, K, dist, seed=None, mutuality=True,
                      theta_prior=(0.1, 0.1), lambda_prior=(10.0, 10.0), eta_prior=(0.5, 1.0), num_realisations=1,
                      max_iter=500, convergence_tol=0.1, decision=1, bias0=0.0, EPS=1e-12, device=None):
    """X_local/R_local: [len(local_layers), N, N, M] slices owned by this rank (numpy or torch-on-GPU uint8).
    Every rank passes the same seed and hyper-parameters.  Returns {"trace", "maxL", "posterior" (local layers)}."""
    local_layers = list(local_layers)
    eng = CaviEngine(X_local, R_local, K=K, mutuality=mutuality, eps=EPS, device=device)
    red_dev = "cpu" if dist.get_backend() == "gloo" else f"cuda:{eng.device}"
    on_dev = red_dev != "cpu"
    if on_dev:
        import torch
        ext = torch.cuda.ExternalStream(eng.stream_ptr(), device=torch.device(red_dev))
        with torch.cuda.stream(ext):   # (filled on the stream every later use of it is queued on: the engine's own)
            buf = torch.zeros(3, dtype=torch.float64, device=red_dev)
    Ll, N, M = eng.L, eng.N, eng.M
    assert Ll == len(local_layers)
    try:
        sum_local, cov = eng.data_stats()
        sum_x = _allreduce([sum_local], dist, red_dev)[0]
        a_th, b_th = theta_prior
        a_la, b_la = lambda_prior
        a_eta, b_eta = eta_prior
        eng.set_priors(a_th, b_th, a_la, b_la, a_eta, b_eta)
        maxL, best, trace = -INF, None, []
        prng = np.random.RandomState(seed)
        for r in range(num_realisations):
            bias = 0.0 if r < 5 else (r - 4) * bias0
            # RandomState stream of the reference (model.py:470, 570-592): rand(L,N,N,K) is drawn layer by layer; this rank
            # needs its own layers only and skips the generator over the others
            pr = None
            if local_layers == sorted(set(local_layers)):
                pr = _hostlib.draw_pr_rho_layers(prng, L_total, N, K, bias, local_layers, cov)
            if pr is None:   # no helper (or layers out of order): draw everything, keep ours
                pr = np.empty((Ll, N, N, K))
                for l in range(L_total):
                    blk = prng.rand(N, N, K)
                    if l in local_layers:
                        pr[local_layers.index(l)] = 1.0 + 0.01 * blk
                pr[..., 0] += bias
                pr /= pr.sum(axis=-1)[..., None]
                onehot = np.zeros(K)
                onehot[0] = 1.0
                pr[cov == 0] = onehot
            gs = (a_th * prng.random_sample((L_total, M)) + a_th)[local_layers]
            ps = (a_la * prng.random_sample((L_total, K)) + a_la)[local_layers]
            gr = (b_th * prng.random_sample((L_total, M)) + b_th)[local_layers]
            prt = (b_la * prng.random_sample((L_total, K)) + b_la)[local_layers]
            if mutuality:
                nu_shp, nu_rte = a_eta * prng.random_sample(1)[0] + a_eta, b_eta + sum_x
            else:
                nu_shp, nu_rte = 1e-6, 1.0
            eng.set_state(gs, gr, ps, prt, nu_shp, nu_rte, pr)
            del pr
            coincide, it, reached, elbo = 0, 1, False, -INF
            while not reached and it <= max_iter:
                check = it == 1 or it % 10 == 0 or it == max_iter
                t0 = time.time()
                if on_dev:
                    # the exchange stays on the device and on the engine's stream: sweep -> all-reduce -> commit are queued
                    # back to back, nothing waits on the host until an ELBO is due
                    eng.sweep_local_dev(buf, want_elbo=check)
                    with torch.cuda.stream(ext):
                        dist.all_reduce(buf)
                    if mutuality:
                        eng.commit_nu_dev(buf)
                    if check:
                        with torch.cuda.stream(ext):
                            tot = buf.cpu().tolist()   # (synchronises the engine's stream)
                        if mutuality:
                            nu_shp = a_eta + tot[0]
                else:
                    parts = eng.sweep_local(want_elbo=check)
                    tot = _allreduce(list(parts) if check else [parts[0]], dist, red_dev)
                    if mutuality:
                        eng.commit_nu(tot[0])
                        nu_shp = a_eta + tot[0]
                if check:
                    old = elbo
                    elbo = tot[1] - (nu_shp / nu_rte) * tot[2] + float(_gamma_term(a_eta, b_eta, nu_shp, nu_rte))
                    if np.isnan(elbo):
                        raise ValueError("ELBO is NaN!!!!")
                    coincide = coincide + 1 if abs(elbo - old) < convergence_tol else 0
                runtime = time.time() - t0
                if coincide > decision:
                    reached = True
                it += 1
                if (it - 1) % 10 == 0:
                    trace.append((r, seed, it - 1, elbo, runtime, reached))
            if maxL < elbo:
                maxL, best = elbo, eng.get_state(rho=True)
            step = prng.randint(1, 500)
            seed = step if seed is None else seed + step
            prng = np.random.RandomState(seed)
        return {"trace": trace, "maxL": maxL, "posterior": best, "layers": local_layers, "next_seed": seed}
    finally:
        eng.close()
