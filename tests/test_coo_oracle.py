"""Pin the coordinate-list oracle (oracle/cavi_coo.c) to the golden vectors dumped from the reference, and to the
dense C oracle on a seeded network with mixed mask rows."""
import numpy as np
import pytest

from oracle import cavi_coo, cavi_ref
from oracle import vimure_oracle as vo
from tests.golden_util import case_config, case_names, load_case


def _coo(d, st, pr, K, mut, R="coo"):
    X = d["X"]
    Rarg = None if R is None else d["R"]
    return cavi_coo.CooRef(X, Rarg, X.shape, K, mut,
                           (pr.alpha_theta, pr.beta_theta, pr.alpha_lambda, pr.beta_lambda, pr.alpha_eta, pr.beta_eta),
                           st.gamma_shp, st.gamma_rte, st.phi_shp, st.phi_rte, st.nu_shp, st.nu_rte, st.pr_rho)


@pytest.mark.parametrize("name", case_names())
def test_coo_oracle_substeps(name):
    d = load_case(name)
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    L, N, _, M = d["X"].shape
    pr = vo.make_priors(L, M, K, **priors)
    pb = vo.Problem(d["X"], d["R"], K, mut, pr, undirected=und)
    st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
    all_ones = bool(np.all(d["R"] == 1))
    for rmode in (["coo", None] if all_ones else ["coo"]):   # an all-ones mask also through the implicit form
        c = _coo(d, st, pr, K, mut, rmode)
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


@pytest.mark.parametrize("mut", [True, False])
def test_coo_oracle_equals_dense_c_oracle(mut):
    from vimure_amd.synthetic import standard_sbm
    L, N, M, K = 2, 40, 30, 3
    net = standard_sbm(N=N, M=M, L=L, K=K, avg_degree=5.0, eta=0.4 if mut else 0.0, seed=5)
    g = np.random.RandomState(3)
    kind = g.randint(0, 3, size=(L, N, N, 1))
    R = np.where(kind == 0, 1, np.where(kind == 1, 0, g.rand(L, N, N, M) < 0.5)).astype(np.uint8)
    pr = 1.0 + 0.01 * g.rand(L, N, N, K)
    pr /= pr.sum(-1)[..., None]
    init = (0.1 + 0.1 * g.rand(L, M), 0.1 + 0.1 * g.rand(L, M), 10 + 10 * g.rand(L, K), 10 + 10 * g.rand(L, K),
            0.7 if mut else 1e-6, 1.0 + float(net.X.sum()) if mut else 1.0, pr)
    pri = (0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    a = cavi_ref.CRef(net.X, R, K, mut, pri, *init)
    b = cavi_coo.CooRef(net.X, R, net.X.shape, K, mut, pri, *init)
    for _ in range(4):
        a.cavi_step()
        b.cavi_step()
    np.testing.assert_allclose(b.rho, a.rho, rtol=1e-9, atol=1e-15)
    for n in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte"):
        np.testing.assert_allclose(getattr(b, n), getattr(a, n), rtol=1e-10)
    assert abs(b.nu_shp - a.nu_shp) <= 1e-11 * abs(a.nu_shp)
    assert abs(b.elbo() - a.elbo()) <= 1e-11 * abs(a.elbo())
