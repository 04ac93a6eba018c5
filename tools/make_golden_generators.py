#!/usr/bin/env python3
"""Generate tests/golden/K_generators.npz from the REAL reference's synthetic generators (build container only).

Imports /root/reference/src/python/vimure with the container-only stubs of tools/oracle_stubs (sktensor, igraph) and
records, for small seeded cases, what the reference's `StandardSBM`, `DegreeCorrectedSBM`, `Multitensor`, `_build_X`
(self-reporter and all-ones masks, union / intersection baselines), `build_self_reporter_mask` and `build_custom_theta`
produce.  The reference never travels: only these data vectors are committed.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_generators.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src/python")
sys.path.insert(0, os.path.join(HERE, "oracle_stubs"))
warnings.filterwarnings("ignore")

import vimure as vm  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "K_generators.npz")


def dense(t, shape=None):
    if t is None:
        return np.zeros(0)
    return np.asarray(t.toarray()) if hasattr(t, "toarray") else np.asarray(t)


def main():
    out = {}
    cases = {
        "sbm": (vm.synthetic.StandardSBM, dict(N=24, M=24, L=2, K=3, C=3, avg_degree=4, sparsify=True, seed=5,
                                               structure=["assortative", "disassortative"])),
        "sbm_nosparse": (vm.synthetic.StandardSBM, dict(N=20, M=6, L=1, K=2, C=2, avg_degree=3, sparsify=False, seed=1)),
        "dcsbm": (vm.synthetic.DegreeCorrectedSBM, dict(N=30, M=30, L=1, K=2, C=2, avg_degree=3, sparsify=True, seed=7,
                                                        exp_in=2, exp_out=2.5)),
        "multitensor": (vm.synthetic.Multitensor, dict(N=26, M=26, L=2, K=2, C=2, avg_degree=5, sparsify=True, seed=25, eta=0.2)),
    }
    build = {
        "sbm": dict(mutuality=0.4, flag_self_reporter=True, seed=11),
        "sbm_nosparse": dict(mutuality=0.3, flag_self_reporter=False, seed=3, cutoff_X=True),
        "dcsbm": dict(mutuality=0.0, flag_self_reporter=True, lambda_diff=0.8, seed=2),
        "multitensor": dict(mutuality=0.2, flag_self_reporter=True, seed=25),
    }
    for name, (cls, kw) in cases.items():
        net = cls(**kw)
        for k, v in kw.items():
            if k != "structure":
                out[f"{name}_arg_{k}"] = np.asarray(v)
        if "structure" in kw:
            out[f"{name}_structure"] = np.asarray(kw["structure"])
        out[f"{name}_Y"] = dense(net.Y)
        out[f"{name}_u"], out[f"{name}_v"], out[f"{name}_w"] = np.asarray(net.u), np.asarray(net.v), np.asarray(net.w)
        if name == "dcsbm":
            out[f"{name}_d_in"], out[f"{name}_d_out"] = np.asarray(net.d_in), np.asarray(net.d_out)
        b = dict(build[name])
        if name == "multitensor":   # the reference's F1 tests build theta this way (test/test_model.py:117-188)
            b["theta"] = vm.synthetic.build_custom_theta(net, theta_ratio=0.1, exaggeration_type="over", seed=25)
            out[f"{name}_custom_theta"] = b["theta"]
        net._build_X(**b)
        for k, v in build[name].items():
            out[f"{name}_build_{k}"] = np.asarray(v)
        out[f"{name}_X"], out[f"{name}_R"] = dense(net.X), dense(net.R)
        out[f"{name}_theta"], out[f"{name}_lambda_k"] = np.asarray(net.theta), np.asarray(net.lambda_k)
        out[f"{name}_X_union"] = dense(net.X_union)
        out[f"{name}_has_intersection"] = np.asarray(net.X_intersection is not None)
        out[f"{name}_X_intersection"] = dense(net.X_intersection)
    net = vm.synthetic.StandardSBM(N=12, M=12, L=2, K=2, seed=0)
    out["mask_self"] = np.asarray(vm.synthetic.build_self_reporter_mask(net))
    out["custom_theta_under"] = vm.synthetic.build_custom_theta(net, theta_ratio=0.5, exaggeration_type="under", seed=4)
    out["custom_theta_over"] = vm.synthetic.build_custom_theta(net, theta_ratio=0.25, exaggeration_type="over", seed=9)
    # posterior predictive generator (synthetic.py:964-1177) on a synthetic "fitted model"
    g = np.random.RandomState(3)
    Lp, Np, Kp = 2, 14, 3

    class Fitted:
        pass
    fm = Fitted()
    fm.rho_f = np.where(g.rand(Lp, Np, Np, 1) < 0.12, g.dirichlet(np.ones(Kp) * 0.6, size=(Lp, Np, Np)),
                        np.array([0.985, 0.01, 0.005]))   # a posterior as fits give it: most ties are confidently absent
    fm.gamma_shp_f, fm.gamma_rte_f = 1.0 + 3.0 * g.rand(Lp, Np), 1.0 + 2.0 * g.rand(Lp, Np)
    fm.phi_shp_f, fm.phi_rte_f = np.array([[0.2, 12.0, 30.0], [0.3, 9.0, 22.0]]), np.array([[10.0, 10.0, 12.0], [9.0, 8.0, 11.0]])
    fm.nu_shp_f, fm.nu_rte_f = np.float64(3.0), np.float64(12.0)
    for k in ("rho_f", "gamma_shp_f", "gamma_rte_f", "phi_shp_f", "phi_rte_f", "nu_shp_f", "nu_rte_f"):
        out["post_" + k] = np.asarray(getattr(fm, k))
    pn = vm.synthetic.PosteriorSyntheticNetwork(fm, seed_Y=7)
    pn.build_Y()
    out["post_Y"] = dense(pn.Y)
    pn.build_X(Rinput=None, flag_self_reporter=True, seed_X=5, verbose=False)
    out["post_X"], out["post_R"] = dense(pn.X), dense(pn.R)
    out["post_theta"], out["post_lambda_k"], out["post_mutuality"] = pn.theta, pn.lambda_k, np.asarray(pn.mutuality)
    out["post_lambda_aux"] = pn.lambda_k_auxiliary
    pn.build_X(Rinput=None, flag_self_reporter=False, cutoff_X=True, seed_X=6, verbose=False)
    out["post_X_ones"] = dense(pn.X)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if v.ndim > 1})


if __name__ == "__main__":
    main()
