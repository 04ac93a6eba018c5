"""Pin the CPU oracle (oracle/vimure_oracle.py) to vectors dumped from the real reference.

Every golden case: RandomState-exact initial state, each CAVI sub-step of the first
iterations, the ELBO after each of them, and a complete fit (stop iteration, trace, maxL,
final posteriors).  Also the reference's own known answers: F1 ~ 0.92 / 0.97
(reference test/test_model.py:117-334) and BASELINE config 1.
"""
import numpy as np
import pytest

from oracle import vimure_oracle as vo
from tests.golden_util import case_config, case_names, load_case

RTOL = 1e-9


def build(d):
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    L, N, _, M = d["X"].shape
    pr = vo.make_priors(L, M, K, **priors)
    pb = vo.Problem(d["X"], d["R"], K, mut, pr, undirected=und)
    return pb, seed, fitargs, rho_prior


@pytest.mark.parametrize("name", case_names())
def test_init_is_bit_exact(name):
    d = load_case(name)
    pb, seed, _, rho_prior = build(d)
    st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
    for n in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "pr_rho"):
        assert np.array_equal(getattr(st, n), d["init_" + n]), n
    assert st.nu_shp == float(d["init_nu_shp"]) and st.nu_rte == float(d["init_nu_rte"])


@pytest.mark.parametrize("name", case_names())
def test_substeps_and_elbo(name):
    d = load_case(name)
    pb, seed, _, rho_prior = build(d)
    st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
    for it in range(1, len(d["step_elbo"]) + 1):
        vo.update_gamma(pb, st)
        np.testing.assert_allclose(st.gamma_shp, d[f"it{it}_gamma_shp"], rtol=RTOL)
        np.testing.assert_allclose(st.gamma_rte, d[f"it{it}_gamma_rte"], rtol=RTOL)
        vo.update_phi(pb, st)
        np.testing.assert_allclose(st.phi_shp, d[f"it{it}_phi_shp"], rtol=RTOL)
        np.testing.assert_allclose(st.phi_rte, d[f"it{it}_phi_rte"], rtol=RTOL)
        vo.update_rho(pb, st)
        np.testing.assert_allclose(st.rho, d[f"it{it}_rho"], rtol=RTOL, atol=1e-300)
        vo.update_nu(pb, st)
        if pb.mutuality:
            np.testing.assert_allclose(st.nu_shp, d[f"it{it}_nu_shp"], rtol=RTOL)
        ref = d["step_elbo"][it - 1]
        assert abs(vo.elbo(pb, st) - ref) <= 1e-9 * max(1.0, abs(ref))


@pytest.mark.parametrize("name", case_names())
def test_full_fit(name):
    d = load_case(name)
    pb, seed, fitargs, rho_prior = build(d)
    res = vo.fit(pb, seed=seed, rho_prior=rho_prior, **fitargs)
    assert [t[2] for t in res.trace] == d["fit_trace_iter"].tolist()
    assert [t[1] for t in res.trace] == d["fit_trace_seed"].tolist()
    assert [int(t[4]) for t in res.trace] == d["fit_trace_conv"].tolist()
    np.testing.assert_allclose([t[3] for t in res.trace], d["fit_trace_elbo"], rtol=1e-9)
    assert abs(res.maxL - float(d["fit_maxL"])) <= 1e-9 * max(1.0, abs(float(d["fit_maxL"])))
    assert res.next_seed == int(d["fit_final_seed"])
    st = res.best
    np.testing.assert_allclose(st.rho, d["fit_rho_f"], rtol=1e-7, atol=1e-12)
    for n in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte"):
        np.testing.assert_allclose(getattr(st, n), d["fit_" + n + "_f"], rtol=1e-8)
    np.testing.assert_allclose(st.nu_shp, d["fit_nu_shp_f"], rtol=1e-8)
    g_th, g_la, g_nu = vo.geometric_means(st)
    np.testing.assert_allclose(g_th, d["fit_G_exp_theta_f"], rtol=1e-8)
    np.testing.assert_allclose(g_la, d["fit_G_exp_lambda_f"], rtol=1e-8)
    np.testing.assert_allclose(g_nu, d["fit_G_exp_nu_f"], rtol=1e-8)


@pytest.mark.parametrize("tag,expected", [("over", 0.92), ("under", 0.97)])
def test_reference_known_answer_f1(tag, expected):
    """reference test/test_model.py:117-188 (over) and :263-334 (under): F1(Y_true, rho>=0.5) ~ expected +- 0.01."""
    from sklearn.metrics import f1_score
    d = load_case(f"H_ref_f1_{tag}")
    pb, seed, fitargs, rho_prior = build(d)
    res = vo.fit(pb, seed=seed, **fitargs)
    y_rec = (res.best.rho[0, :, :, 1] >= 0.5).astype(int)
    f1 = f1_score(d["Y_true"].flatten(), y_rec.flatten())
    assert abs(f1 - expected) <= 1e-2
    assert abs(f1 - float(d["f1"])) < 1e-12


def test_config1_known_values():
    """SURVEY App. B, C1 plumbing: StandardSBM N=100 M=10, fit(seed=1) converges at iter 50."""
    d = load_case("G_config1_sbm")
    pb, seed, fitargs, _ = build(d)
    res = vo.fit(pb, seed=seed, **fitargs)
    assert res.trace[-1][2] == 50 and res.trace[-1][4]
    assert abs(res.maxL - (-31.619036359447087)) < 1e-8
