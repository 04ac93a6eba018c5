#!/usr/bin/env python3
"""Print (not assert) per-sub-step errors of the HIP engine vs the oracle for every golden case.
Used during bring-up so that one GPU call reports everything at once."""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vimure_oracle as vo  # noqa: E402
from tests.golden_util import case_config, case_names, load_case  # noqa: E402
from vimure_amd import CaviEngine, _lib  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    den = np.maximum(np.abs(b), 1e-300)
    e = np.abs(a - b) / den
    i = int(np.argmax(e))
    return f"max_rel={e.max():.3e} (abs {np.abs(a-b).max():.3e}) at {np.unravel_index(i, e.shape) if e.ndim else ()} got={a.flat[i]!r} want={b.flat[i]!r}"


def main():
    names = sys.argv[1:] or case_names()
    for name in names:
        try:
            d = load_case(name)
            K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
            L, N, _, M = d["X"].shape
            print(f"=== {name}: L={L} N={N} M={M} K={K} mut={mut}", flush=True)
            pr = vo.make_priors(L, M, K, **priors)
            pb = vo.Problem(d["X"], d["R"], K, mut, pr, undirected=und)
            st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
            eng = CaviEngine(d["X"], d["R"], K=K, mutuality=mut)
            s, cov = eng.data_stats()
            print("  sumX", s, pb.sumX, "cov ok", np.array_equal(cov, (pb.R.any(axis=3) & (pb.X != 0).any(axis=3))))
            eng.set_priors(pr.alpha_theta, pr.beta_theta, pr.alpha_lambda, pr.beta_lambda, pr.alpha_eta, pr.beta_eta)
            eng.set_state(st.gamma_shp, st.gamma_rte, st.phi_shp, st.phi_rte, st.nu_shp, st.nu_rte, st.pr_rho)
            g = eng.get_state()
            print("  init rho   ", rel(g["rho"], st.rho))
            print("  elbo@init  ", eng.elbo(), vo.elbo(pb, st))
            for it in range(1, min(2, len(d["step_elbo"])) + 1):
                eng.sub_step(_lib.STEP_GAMMA); vo.update_gamma(pb, st); g = eng.get_state(rho=False)
                print(f"  it{it} gamma_shp", rel(g["gamma_shp"], st.gamma_shp))
                print(f"  it{it} gamma_rte", rel(g["gamma_rte"], st.gamma_rte))
                eng.sub_step(_lib.STEP_PHI); vo.update_phi(pb, st); g = eng.get_state(rho=False)
                print(f"  it{it} phi_shp  ", rel(g["phi_shp"], st.phi_shp))
                print(f"  it{it} phi_rte  ", rel(g["phi_rte"], st.phi_rte))
                eng.sub_step(_lib.STEP_RHO); vo.update_rho(pb, st); g = eng.get_state()
                print(f"  it{it} rho      ", rel(g["rho"], st.rho))
                eng.sub_step(_lib.STEP_NU); vo.update_nu(pb, st); g = eng.get_state(rho=False)
                print(f"  it{it} nu_shp   ", rel(g["nu_shp"], st.nu_shp))
                print(f"  it{it} elbo      gpu={eng.elbo()!r} oracle={vo.elbo(pb, st)!r} golden={float(d['step_elbo'][it-1])!r}")
            eng.close()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()


if __name__ == "__main__":
    main()
