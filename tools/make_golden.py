#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (build container only).

Imports /root/reference/src/python/vimure with the two container-only stub
modules in tools/oracle_stubs (sktensor, igraph: absent offline, containers
only -- SURVEY.md App. A) and records, for a set of small seeded problems:
inputs, the RandomState-seeded initial state, the state after every sub-step
(gamma, phi, rho, nu) of the first iterations, the ELBO after each of those
iterations, and the outputs of a complete `fit` (trace, maxL, *_f arrays).

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
The reference never travels: only these data vectors are committed.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/python"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "oracle_stubs"))
warnings.filterwarnings("ignore")

import pandas as pd  # noqa: E402
import vimure as vm  # noqa: E402
from vimure.model import VimureModel  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def synth(L, N, M, eta, seed, density=0.06, mask="ones", outside=0):
    """Small random multiply-reported network (own generator, not the reference's)."""
    g = np.random.RandomState(seed)
    theta = g.gamma(2.0, 0.5, size=(L, M))
    Y = (g.rand(L, N, N) < density).astype(float)
    for l in range(L):
        np.fill_diagonal(Y[l], 0)
    lam = np.where(Y > 0, 1.0, 0.01)
    MX = theta[:, None, None, :] * lam[..., None]
    MM = (MX + eta * MX.transpose(0, 2, 1, 3)) / (1 - eta * eta)
    first = g.poisson(MM)
    second = g.poisson(MX.transpose(0, 2, 1, 3) + eta * first)
    iu = np.triu(np.ones((N, N), bool), 1)
    X = np.zeros((L, N, N, M), np.int64)
    X[:, iu, :] = first[:, iu, :]
    Xt = np.zeros_like(X)
    Xt[:, iu, :] = second[:, iu, :]
    X = X + Xt.transpose(0, 2, 1, 3)
    if mask == "ones":
        R = np.ones((L, N, N, M), np.int64)
    elif mask == "random":
        R = (g.rand(L, N, N, M) < 0.35).astype(np.int64)
    elif mask == "self":
        assert M == N
        R = np.zeros((L, N, N, M), np.int64)
        reporters = g.rand(N) < 0.7
        for m in np.nonzero(reporters)[0]:
            R[:, m, :, m] = 1
            R[:, :, m, m] = 1
    X = X * (R > 0)
    if outside:  # X non-zeros outside R
        zl, zi, zj, zm = np.nonzero(R == 0)
        pick = g.choice(len(zl), size=min(outside, len(zl)), replace=False)
        X[zl[pick], zi[pick], zj[pick], zm[pick]] = g.randint(1, 4, size=len(pick))
    for l in range(L):
        for m in range(M):
            np.fill_diagonal(X[l, :, :, m], 0)
    return X, R


def snap(m, names=("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "nu_rte", "rho")):
    return {n: np.array(getattr(m, n), dtype=np.float64) for n in names}


def run_case(name, X, R, K, mutuality=True, undirected=False, seed=1, n_steps=3, fit_kwargs=None,
             rho_prior=None, priors=None, save_inputs=True):
    priors = priors or {}
    fit_kwargs = dict(fit_kwargs or {})
    out = {"K": K if K is not None else int(X.max()) + 1, "K_given": int(K is not None), "mutuality": int(mutuality),
           "undirected": int(undirected), "seed": seed}
    if save_inputs:
        assert X.max() <= 255
        out["X"] = X.astype(np.uint8)
        out["R"] = R.astype(np.uint8)
    # ---- step-level capture through the reference's private methods
    m = VimureModel(mutuality=mutuality, undirected=undirected)
    kk = {} if K is None else {"K": K}   # (K omitted: the reference defaults to max(X) + 1, model.py:192-197)
    m._VimureModel__check_fit_params(X=X.copy(), seed=seed, R=R.copy(), rho_prior=rho_prior, **kk, **priors)
    m._set_rho_prior()
    m._initialize_priors()
    m._initialize_old_variables()
    for k_, v in snap(m, ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "nu_rte", "pr_rho")).items():
        out["init_" + k_] = v
    elbos = []
    for it in range(1, n_steps + 1):
        for step, names in (("gamma", ("gamma_shp", "gamma_rte")), ("phi", ("phi_shp", "phi_rte")),
                            ("rho", ("rho",)), ("nu", ("nu_shp",))):
            if step == "nu" and not m.mutuality:
                continue
            m._update_cache(m.X, m.subs_nz, m.data_T_vals)
            getattr(m, "_update_" + step)(m.subs_nz)
            for n in names:
                out[f"it{it}_{n}"] = np.array(getattr(m, n), dtype=np.float64)
        elbos.append(m._VimureModel__ELBO(m.X, m.data_T, m.subs_nz))
    out["step_elbo"] = np.array(elbos)
    # ---- a complete fit
    m2 = VimureModel(mutuality=mutuality, undirected=undirected)
    m2.fit(X.copy(), R=R.copy(), seed=seed, rho_prior=rho_prior, **kk, **priors, **fit_kwargs)
    for n in ("gamma_shp_f", "gamma_rte_f", "phi_shp_f", "phi_rte_f", "nu_shp_f", "nu_rte_f", "rho_f",
              "G_exp_theta_f", "G_exp_lambda_f", "G_exp_nu_f"):
        out["fit_" + n] = np.array(getattr(m2, n), dtype=np.float64)
    out["fit_maxL"] = np.array(m2.maxL)
    out["fit_final_seed"] = np.array(m2.seed)
    tr = m2.trace
    out["fit_trace_realisation"] = tr["realisation"].values.astype(np.int64)
    out["fit_trace_seed"] = tr["seed"].values.astype(np.int64)
    out["fit_trace_iter"] = tr["iter"].values.astype(np.int64)
    out["fit_trace_elbo"] = tr["elbo"].values.astype(np.float64)
    out["fit_trace_conv"] = tr["reached_convergence"].values.astype(np.int64)
    for k_, v in fit_kwargs.items():
        out["fitarg_" + k_] = np.array(v)
    for k_, v in priors.items():
        out["prior_" + k_] = np.array(v)
    if rho_prior is not None:
        out["rho_prior"] = rho_prior
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: nnzX={np.count_nonzero(X)} nnzR={np.count_nonzero(R)} maxL={m2.maxL!r} "
          f"iters={tr['iter'].values.tolist()[-3:]} stepELBO={elbos}")
    return m2


def coo_inputs(X, R):
    xs, rs = np.nonzero(X), np.nonzero(R)
    return {"X_shape": np.array(X.shape), "X_subs": np.stack(xs).astype(np.int16), "X_vals": X[xs].astype(np.uint8),
            "R_subs": np.stack(rs).astype(np.int16)}


def main():
    os.makedirs(OUT, exist_ok=True)
    # A: all-ones dense R, mutuality on
    X, R = synth(1, 30, 8, 0.5, 11)
    run_case("A_ones_mut", X, R, 2, seed=1, fit_kwargs=dict(num_realisations=2, max_iter=40))
    # B: two layers, K=3, random sparse mask with X entries outside R
    X, R = synth(2, 24, 10, 0.4, 12, mask="random", outside=15)
    run_case("B_random_mask_K3", X, R, 3, seed=7, fit_kwargs=dict(num_realisations=2, max_iter=30))
    # C: mutuality off
    X, R = synth(1, 30, 12, 0.0, 13)
    run_case("C_ones_nomut", X, R, 2, mutuality=False, seed=3, fit_kwargs=dict(num_realisations=1, max_iter=40))
    # D: self-reporter mask (Karnataka-like), two layers, M == N
    X, R = synth(2, 20, 20, 0.3, 14, mask="self", density=0.15)
    run_case("D_self_mask", X, R, 2, seed=5, fit_kwargs=dict(num_realisations=6, max_iter=21, bias0=0.2))
    # E: undirected (symmetric X forces mutuality off)
    X, R = synth(1, 26, 9, 0.0, 15)
    X = np.maximum(X, X.transpose(0, 2, 1, 3))
    run_case("E_undirected", X, R, 2, mutuality=True, undirected=True, seed=9,
             fit_kwargs=dict(num_realisations=1, max_iter=30))
    # F: informative rho_prior + array-valued lambda priors
    X, R = synth(1, 22, 10, 0.5, 16)
    rp = (X.sum(axis=3) > 0).astype(float) * 0.8 + 0.1 * np.random.RandomState(3).rand(1, 22, 22)
    rp[0, :3, :3] = 0.0
    run_case("F_rho_prior", X, R, 2, seed=4, rho_prior=rp,
             priors=dict(alpha_lambda=np.array([[100.0, 10000.0]]), beta_lambda=1e4 * np.ones((1, 2))),
             fit_kwargs=dict(num_realisations=1, max_iter=20))
    # G: BASELINE config 1 -- the reference's own StandardSBM generator (SURVEY App. B, C1 plumbing)
    from vimure.synthetic import StandardSBM, Multitensor
    g = StandardSBM(N=100, M=10, L=1, K=2, C=2, avg_degree=2, sparsify=False, seed=0)._build_X(flag_self_reporter=True)
    X, R = g.X.toarray(), g.R.toarray()
    run_case("G_config1_sbm", X, R, 2, seed=1, n_steps=2)
    # H: the reference's own known-answer tests (test/test_model.py:117-334): F1 ~ 0.92 (over) / 0.97 (under)
    from sklearn.metrics import f1_score
    for tag in ("over", "under"):
        eta = 0.2
        gt = Multitensor(N=100, M=100, L=1, C=2, K=2, avg_degree=5, sparsify=True, seed=25, ExpM=None, eta=eta)
        theta = vm.synthetic.build_custom_theta(gt_network=gt, theta_ratio=0.1, exaggeration_type=tag, seed=25)
        gt._build_X(mutuality=eta, theta=theta, cutoff_X=False, lambda_diff=0.99, flag_self_reporter=True, seed=25)
        X, R = gt.X.toarray(), gt.R.toarray()
        m2 = run_case(f"H_ref_f1_{tag}", X, R, 2, seed=25, n_steps=1, save_inputs=False,
                      priors=dict(alpha_lambda=np.array([[100.0, 10000.0]]), beta_lambda=1e4 * np.ones((1, 2))),
                      fit_kwargs=dict(num_realisations=2, max_iter=21))
        Yrec = (m2.rho_f[0, :, :, 1] >= 0.5).astype(int)
        Ytrue = gt.Y.toarray()[0]
        f1 = f1_score(Ytrue.flatten(), Yrec.flatten())
        path = os.path.join(OUT, f"H_ref_f1_{tag}.npz")
        d = dict(np.load(path))
        d.update(coo_inputs(X, R))
        d["Y_true"] = Ytrue.astype(np.uint8)
        d["f1"] = np.array(f1)
        np.savez_compressed(path, **d)
        print(f"  F1({tag}) = {f1:.4f}")
    # I: Karnataka village 1, layer 'money' -- data file the reference's own tests read
    #    (test/__init__.py:18-41, test_model.py:450-481)
    sys.path.insert(0, "/root/reference/notebooks/python/experiments/")
    from karnataka import read_village_data
    df, _, _ = read_village_data("vil1", filter_layer="money", print_details=False,
                                 data_folder="/root/reference/data/input/india_microfinance/formatted/")
    df.rename(columns={"Ego": "ego", "Alter": "alter"}, inplace=True)
    net = vm._io.read_from_edgelist(df, K=2)
    X, R = net.X.toarray(), net.R.toarray()
    run_case("I_karnataka_vil1_money", X, R, 2, seed=1, n_steps=1, save_inputs=False,
             fit_kwargs=dict(num_realisations=1, max_iter=21))
    path = os.path.join(OUT, "I_karnataka_vil1_money.npz")
    d = dict(np.load(path))
    d.update(coo_inputs(X, R))
    np.savez_compressed(path, **d)


def io_cases():
    """Edgelist ingestion vectors (reference _io.py:132-295): a small synthetic survey and Karnataka village 1."""
    g = np.random.RandomState(5)
    names = [f"n{i:02d}" for i in range(14)]
    rows = []
    for rep in names[:10]:
        for _ in range(g.randint(2, 7)):
            other = names[g.randint(len(names))]
            if other == rep:
                continue
            e, a = (rep, other) if g.rand() < 0.5 else (other, rep)
            rows.append((rep, e, a, ["borrow", "advice"][g.randint(2)], int(g.randint(0, 4))))
    df = pd.DataFrame(rows, columns=["reporter", "ego", "alter", "layer", "weight"])
    out = {}
    for tag, kw in (("plain", {}), ("weighted", dict(is_weighted=True)), ("undirected", dict(is_undirected=True, is_weighted=True)),
                    ("lists", dict(nodes=sorted(names), reporters=sorted(names[:10]), K=3))):
        net = vm._io.read_from_edgelist(df.copy(), **kw)
        out[f"{tag}_X_subs"] = np.stack(net.X.subs).astype(np.int32)
        out[f"{tag}_X_vals"] = np.asarray(net.X.vals).astype(np.int64)
        out[f"{tag}_R_subs"] = np.stack(net.R.subs).astype(np.int32)
        out[f"{tag}_shape"] = np.array(net.X.shape)
        out[f"{tag}_LNMK"] = np.array([net.L, net.N, net.M, net.K])
        out[f"{tag}_nodes"] = np.array(net.nodeNames["name"].tolist())
        out[f"{tag}_layers"] = np.array(net.layerNames)
    for c in df.columns:
        out["df_" + c] = df[c].values.astype(str) if df[c].dtype == object else df[c].values
    sys.path.insert(0, "/root/reference/notebooks/python/experiments/")
    from karnataka import read_village_data
    kdf, _, _ = read_village_data("vil1", filter_layer="money", print_details=False,
                                  data_folder="/root/reference/data/input/india_microfinance/formatted/")
    kdf.rename(columns={"Ego": "ego", "Alter": "alter"}, inplace=True)
    for c in kdf.columns:
        out["vil1_df_" + c] = kdf[c].values.astype(str) if kdf[c].dtype == object else kdf[c].values
    net = vm._io.read_from_edgelist(kdf.copy(), K=2)
    out["vil1_LNMK"] = np.array([net.L, net.N, net.M, net.K])
    out["vil1_nodes"] = np.array(net.nodeNames["name"].tolist())
    np.savez_compressed(os.path.join(OUT, "J_edgelist_io.npz"), **out)
    print("J_edgelist_io: rows", len(df), "vil1 rows", len(kdf), "vil1 LNMK", out["vil1_LNMK"])


def wide_cases():
    """Inputs beyond K = 8 and beyond one-byte counts: the reference's DEFAULT K = max(X) + 1 (model.py:179-197, exercised by
    test/test_model.py:59-115) and counts in the thousands (edgelist weights summed per (reporter, tie), utils.py:241-242)."""
    # L: K omitted -> K = max(X) + 1 = 12; random mask, mutuality on.  Counts 1..11 on top of a sparse base.
    X, R = synth(1, 18, 8, 0.4, 21, mask="random", density=0.10)
    g = np.random.RandomState(77)
    nz = np.nonzero(X)
    pick = g.choice(len(nz[0]), size=len(nz[0]) // 3, replace=False)
    X[nz[0][pick], nz[1][pick], nz[2][pick], nz[3][pick]] += g.randint(1, 11, size=len(pick))
    X = np.minimum(X, 11)
    X[0, 2, 3, 1] = 11
    R[0, 2, 3, 1] = 1
    run_case("L_default_K12", X, R, None, seed=2, n_steps=2, fit_kwargs=dict(num_realisations=2, max_iter=30))
    # M: two layers, K = 16 given, all-ones mask, mutuality off
    X, R = synth(2, 14, 6, 0.0, 22, density=0.12)
    X = X * g.randint(1, 4, size=X.shape)
    run_case("M_K16_nomut", X, R, 16, mutuality=False, seed=4, n_steps=2, fit_kwargs=dict(num_realisations=1, max_iter=20))
    # N: K = 2 with counts up to 12000 (int64 in the reference, utils.py:241-242), self-reporter mask.  The reference accepts the
    #    tensor and then fails on its own arithmetic: x (E log theta + E log lambda) passes 709, the raw exponentials of the rho
    #    update (model.py:807) overflow, inf / inf = NaN, and the first ELBO check raises "ELBO is NaN!!!!" (model.py:1016).
    #    Recorded: the inputs and the error, which the engine has to reproduce (it accepts the counts, then raises the same).
    X, R = synth(1, 16, 16, 0.3, 23, mask="self", density=0.2, outside=6)
    nz = np.nonzero(X)
    pick = g.choice(len(nz[0]), size=len(nz[0]) // 2, replace=False)
    X[nz[0][pick], nz[1][pick], nz[2][pick], nz[3][pick]] *= g.randint(2, 4000, size=len(pick))
    X[0, 1, 2, 1] = 12000
    R[0, 1, 2, 1] = 1
    err = ""
    try:
        VimureModel().fit(X.copy(), R=R.copy(), K=2, seed=6, num_realisations=1, max_iter=20)
    except ValueError as e:
        err = str(e)
    xs, rs = np.nonzero(X), np.nonzero(R)
    np.savez_compressed(os.path.join(OUT, "N_counts_12000.npz"), K=2, seed=6, error=np.array(err), X_shape=np.array(X.shape),
                        X_subs=np.stack(xs).astype(np.int16), X_vals=X[xs].astype(np.int32), R_subs=np.stack(rs).astype(np.int16))
    print(f"N_counts_12000: nnzX={len(xs[0])} max={X.max()} reference raised: {err!r}")
    # O: many reporters and counts to 130: (max + 1) * M passes the 2^20 table rows of a packed entry; mutuality on, all finite
    #    (a count of 250 already overflows the reference's exponentials)
    L_, N_, M_ = 1, 5, 8190
    g2 = np.random.RandomState(31)
    X = np.zeros((L_, N_, N_, M_), np.int64)
    R = np.ones_like(X)
    for _ in range(160):
        i, j = g2.randint(N_, size=2)
        if i != j:
            X[0, i, j, g2.randint(M_)] = g2.randint(1, 4)
    X[0, 1, 2, 7] = 130
    X[0, 2, 1, 7] = 3
    X[0, 3, 4, 8189] = 60
    run_case("O_wide_rows", X, R, 2, seed=8, n_steps=2, save_inputs=False, fit_kwargs=dict(num_realisations=1, max_iter=20))
    path = os.path.join(OUT, "O_wide_rows.npz")
    d = dict(np.load(path))
    xs = np.nonzero(X)
    d.update({"X_shape": np.array(X.shape), "X_subs": np.stack(xs).astype(np.int16), "X_vals": X[xs].astype(np.int32)})
    np.savez_compressed(path, **d)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "io":
    io_cases()

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "wide":
    os.makedirs(OUT, exist_ok=True)
    wide_cases()


if __name__ == "__main__" and len(sys.argv) == 1:
    main()
    io_cases()
    wide_cases()
