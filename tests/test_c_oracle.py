"""Pin the plain-C oracle (oracle/cavi_ref.c) to the golden vectors dumped from the reference."""
import numpy as np
import pytest

from oracle import cavi_ref
from oracle import vimure_oracle as vo
from tests.golden_util import case_config, case_names, load_case


@pytest.mark.parametrize("name", [n for n in case_names() if not n.startswith("I_")])
def test_c_oracle_substeps(name):
    d = load_case(name)
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    L, N, _, M = d["X"].shape
    pr = vo.make_priors(L, M, K, **priors)
    pb = vo.Problem(d["X"], d["R"], K, mut, pr, undirected=und)
    st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
    c = cavi_ref.CRef(d["X"], d["R"], K, mut,
                      (pr.alpha_theta, pr.beta_theta, pr.alpha_lambda, pr.beta_lambda, pr.alpha_eta, pr.beta_eta),
                      st.gamma_shp, st.gamma_rte, st.phi_shp, st.phi_rte, st.nu_shp, st.nu_rte, st.pr_rho)
    for it in range(1, len(d["step_elbo"]) + 1):
        c.update_gamma()
        np.testing.assert_allclose(c.gamma_shp, d[f"it{it}_gamma_shp"], rtol=1e-9)
        np.testing.assert_allclose(c.gamma_rte, d[f"it{it}_gamma_rte"], rtol=1e-9)
        c.update_phi()
        np.testing.assert_allclose(c.phi_shp, d[f"it{it}_phi_shp"], rtol=1e-9)
        np.testing.assert_allclose(c.phi_rte, d[f"it{it}_phi_rte"], rtol=1e-9)
        c.update_rho()
        np.testing.assert_allclose(c.rho, d[f"it{it}_rho"], rtol=1e-9, atol=1e-13)
        c.update_nu()
        if mut:
            np.testing.assert_allclose(c.nu_shp, d[f"it{it}_nu_shp"], rtol=1e-9)
        ref = float(d["step_elbo"][it - 1])
        assert abs(c.elbo() - ref) <= 1e-9 * max(1.0, abs(ref))


def test_c_oracle_fused_step_matches_numpy_oracle():
    d = load_case("B_random_mask_K3")
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    L, N, _, M = d["X"].shape
    pr = vo.make_priors(L, M, K, **priors)
    pb = vo.Problem(d["X"], d["R"], K, mut, pr)
    st = vo.init_state(pb, np.random.RandomState(seed))
    c = cavi_ref.CRef(d["X"], d["R"], K, mut,
                      (pr.alpha_theta, pr.beta_theta, pr.alpha_lambda, pr.beta_lambda, pr.alpha_eta, pr.beta_eta),
                      st.gamma_shp, st.gamma_rte, st.phi_shp, st.phi_rte, st.nu_shp, st.nu_rte, st.pr_rho)
    for _ in range(5):
        c.cavi_step()
        vo.cavi_step(pb, st)
    np.testing.assert_allclose(c.rho, st.rho, rtol=1e-8, atol=1e-13)
    assert abs(c.elbo() - vo.elbo(pb, st)) < 1e-8
