"""GPU: VMR_DETERMINISTIC=1 -- two runs of the same fit leave bit-identical states and ELBOs (the reference, single-threaded
NumPy, is bit-reproducible: model.py:623-660), and they agree with the default mode within its usual tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(case, sweeps, monkeypatch, det):
    from oracle import vimure_oracle as vo
    from tests.golden_util import case_config, load_case
    from vimure_amd import CaviEngine
    if det:
        monkeypatch.setenv("VMR_DETERMINISTIC", "1")
    else:
        monkeypatch.delenv("VMR_DETERMINISTIC", raising=False)
    d = load_case(case)
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    L, N, _, M = d["X"].shape
    pr = vo.make_priors(L, M, K, **priors)
    pb = vo.Problem(d["X"], d["R"], K, mut, pr, undirected=und)
    st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
    eng = CaviEngine(d["X"], d["R"], K=K, mutuality=mut)
    eng.set_priors(pr.alpha_theta, pr.beta_theta, pr.alpha_lambda, pr.beta_lambda, pr.alpha_eta, pr.beta_eta)
    eng.set_state(st.gamma_shp, st.gamma_rte, st.phi_shp, st.phi_rte, st.nu_shp, st.nu_rte, st.pr_rho)
    elbos = [eng.step(1, want_elbo=True)]
    rows, elbo, its, conv = eng.fit_loop(sweeps, 1e-12, 100)
    elbos += [r[1] for r in rows] + [elbo]
    out = eng.get_state(rho=True)
    eng.close()
    return elbos, out


@pytest.mark.parametrize("case", ["B_random_mask_K3", "D_self_mask", "A_ones_mut", "C_ones_nomut"])
def test_two_deterministic_runs_are_bit_equal(case, monkeypatch):
    import os
    from tests.golden_util import GOLDEN
    if not os.path.exists(os.path.join(GOLDEN, case + ".npz")):
        pytest.skip("no such golden case")
    e1, s1 = _run(case, 31, monkeypatch, True)
    e2, s2 = _run(case, 31, monkeypatch, True)
    assert e1 == e2
    for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "nu_rte", "rho"):
        assert np.array_equal(np.asarray(s1[k]), np.asarray(s2[k])), k
    e0, s0 = _run(case, 31, monkeypatch, False)
    np.testing.assert_allclose(e1, e0, rtol=1e-9)
    for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "rho"):
        np.testing.assert_allclose(np.asarray(s1[k]), np.asarray(s0[k]), rtol=1e-8, atol=1e-11, err_msg=k)


def test_deterministic_config3_sized_run_is_bit_equal(monkeypatch):
    """A network large enough for hundreds of workgroups (L = 2, N = 600, M = 60, mutuality on): the sums that cross workgroups."""
    import torch
    from bench import draw_state
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    monkeypatch.setenv("VMR_DETERMINISTIC", "1")
    net = standard_sbm(N=600, M=60, L=2, K=2, avg_degree=6.0, eta=0.5, seed=3, device="cuda:0")
    outs = []
    for _ in range(2):
        eng = CaviEngine(net.X, None, K=2, mutuality=True, device=0)
        sum_x, cov = eng.data_stats()
        host, pr = draw_state(dict(L=2, N=600, M=60, K=2, mutuality=True), 5, sum_x, cov)
        eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
        eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
        e = eng.step(12, want_elbo=True)
        st = eng.get_state(rho=True)
        outs.append((e, st))
        eng.close()
    assert outs[0][0] == outs[1][0]
    for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "rho"):
        assert np.array_equal(np.asarray(outs[0][1][k]), np.asarray(outs[1][1][k])), k
    del net
    torch.cuda.empty_cache()


def test_deterministic_two_pass_regime_is_bit_equal(monkeypatch):
    """The config-5 regime (M = 1000 reporters, K = 3: a rho pass and a statistics pass per sweep, levels beyond the LDS copies)."""
    import torch
    from bench import draw_state
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    monkeypatch.setenv("VMR_DETERMINISTIC", "1")
    monkeypatch.setenv("VMR_TWO_PASS", "1")
    monkeypatch.setenv("VMR_YT", "2")
    monkeypatch.setenv("VMR_HC", "2")
    net = standard_sbm(N=300, M=1000, L=1, K=3, C=2, avg_degree=5.0, eta=0.5, seed=2, device="cuda:0")
    outs = []
    for _ in range(2):
        eng = CaviEngine(net.X, None, K=3, mutuality=True, device=0)
        sum_x, cov = eng.data_stats()
        host, pr = draw_state(dict(L=1, N=300, M=1000, K=3, mutuality=True), 4, sum_x, cov)
        eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
        eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
        e = [eng.step(1, want_elbo=True) for _ in range(4)]
        outs.append((e, eng.get_state(rho=True)))
        eng.close()
    assert outs[0][0] == outs[1][0]
    for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "rho"):
        assert np.array_equal(np.asarray(outs[0][1][k]), np.asarray(outs[1][1][k])), k
    del net
    torch.cuda.empty_cache()


@pytest.mark.parametrize("case", ["B_random_mask_K3", "D_self_mask", "A_ones_mut"])
def test_deterministic_mode_against_the_oracle_and_the_reference(case, monkeypatch):
    """The deterministic mode against something independent of the engine: the NumPy oracle's sweeps from the same state, and the
    reference's own golden values (sub-step rho / ELBO of iteration 1..3) -- not only against itself and the default mode."""
    from oracle import vimure_oracle as vo
    from tests.golden_util import case_config, load_case
    from vimure_amd import CaviEngine
    monkeypatch.setenv("VMR_DETERMINISTIC", "1")
    d = load_case(case)
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    L, N, _, M = d["X"].shape
    pr = vo.make_priors(L, M, K, **priors)
    pb = vo.Problem(d["X"], d["R"], K, mut, pr, undirected=und)
    st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
    eng = CaviEngine(d["X"], d["R"], K=K, mutuality=mut)
    eng.set_priors(pr.alpha_theta, pr.beta_theta, pr.alpha_lambda, pr.beta_lambda, pr.alpha_eta, pr.beta_eta)
    eng.set_state(st.gamma_shp, st.gamma_rte, st.phi_shp, st.phi_rte, st.nu_shp, st.nu_rte, st.pr_rho)
    for it in range(1, len(d["step_elbo"]) + 1):
        e = eng.step(1, want_elbo=True)
        vo.cavi_step(pb, st)
        eo, er = vo.elbo(pb, st), float(d["step_elbo"][it - 1])
        assert abs(e - eo) <= 1e-9 * max(1.0, abs(eo)) and abs(e - er) <= 1e-9 * max(1.0, abs(er)), (it, e, eo, er)
        g = eng.get_state()
        np.testing.assert_allclose(g["rho"], d[f"it{it}_rho"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(g["gamma_shp"], d[f"it{it}_gamma_shp"], rtol=1e-9)
        np.testing.assert_allclose(g["phi_rte"], d[f"it{it}_phi_rte"], rtol=1e-9)
        if mut:
            np.testing.assert_allclose(g["nu_shp"], d[f"it{it}_nu_shp"], rtol=1e-9)
    eng.close()
