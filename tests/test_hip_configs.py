"""GPU parity at the sizes BASELINE.json states (configs[1], configs[2]) and in the config-5 regime (M = 1000, K = 3),
against the CPU oracles on the same seeded inputs and initial state:

  configs[1]  L=1 N=500 M=50 K=2, mutuality off, R = None and R = ones, both data layouts   -> oracle/cavi_ref.c
  configs[2]  L=4 N=2000 M=200 K=2, mutuality on, full fit to the reference's stop rule       -> oracle/cavi_coo.c
  configs[4]  one layer N=1500 M=1000 K=3 in each mode the engine has for wide reporter dimensions -> oracle/cavi_coo.c

The coordinate-list oracle is used where a dense [L,N,N,M] host tensor is out of reach; it is pinned to the
reference's golden vectors and to the dense oracle in tests/test_coo_oracle.py."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRI = (0.1, 0.1, 10.0, 10.0, 0.5, 1.0)


def _host_state(L, N, M, K, mutuality, seed, sum_x, coverage):
    """RandomState-exact initial state through the host class (reference model.py:458-605)."""
    from vimure_amd.model import VimureModel
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VimureModel(mutuality=mutuality)
    m.L, m.N, m.M, m.K = L, N, M, K
    m.alpha_theta, m.beta_theta, m.alpha_lambda, m.beta_lambda, m.alpha_mutuality, m.beta_mutuality = PRI
    m.rho_prior = None
    m._change_seed(seed)
    pr = m._draw_pr_rho(coverage, 0.0)
    m._draw_gammas(sum_x)
    return (m.gamma_shp, m.gamma_rte, m.phi_shp, m.phi_rte, m.nu_shp, m.nu_rte, pr)


def _coo_from_device(X):
    """Non-zero counts of a device tensor [L,N,N,M] as host (subs, vals), one layer and row block at a time."""
    import torch
    L, N, _, M = X.shape
    subs, vals = [[], [], [], []], []
    step = max(1, int(2.5e8 // (N * M)))
    for l in range(L):
        for i0 in range(0, N, step):
            blk = X[l, i0:i0 + step]
            nz = torch.nonzero(blk, as_tuple=True)
            subs[0].append(np.full(len(nz[0]), l, np.int64))
            subs[1].append((nz[0] + i0).cpu().numpy())
            subs[2].append(nz[1].cpu().numpy())
            subs[3].append(nz[2].cpu().numpy())
            vals.append(blk[nz].cpu().numpy())
    return tuple(np.concatenate(a) for a in subs), np.concatenate(vals)


def _assert_state(st, c, elbo_gpu, elbo_cpu, rho_rtol=1e-7):
    assert abs(elbo_gpu - elbo_cpu) <= 1e-9 * max(1.0, abs(elbo_cpu)), (elbo_gpu, elbo_cpu)
    np.testing.assert_allclose(st["rho"], c.rho, rtol=rho_rtol, atol=1e-12)
    for n in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte"):
        np.testing.assert_allclose(st[n], getattr(c, n), rtol=1e-9, err_msg=n)
    np.testing.assert_allclose(st["nu_shp"], c.nu_shp, rtol=1e-9)


@pytest.mark.parametrize("mask", ["none", "ones"])
def test_config2_full_size_mutuality_off(mask, vmr_format):
    """BASELINE configs[1] at its stated size; the reference needs 2.09 s per sweep + 20.8 s per ELBO here."""
    from oracle import cavi_ref
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    L, N, M, K = 1, 500, 50, 2
    net = standard_sbm(N=N, M=M, L=L, K=K, C=2, avg_degree=5.0, sparsify=True, eta=0.0, seed=0)
    R = None if mask == "none" else np.ones((L, N, N, M), np.uint8)
    eng = CaviEngine(net.X, R, K=K, mutuality=False)
    assert eng.data_format()[0] == vmr_format
    sum_x, cov = eng.data_stats()
    init = _host_state(L, N, M, K, False, 1, sum_x, cov)
    c = cavi_ref.CRef(net.X, R, K, False, PRI, *init)
    eng.set_priors(*PRI)
    eng.set_state(*init)
    for it in range(1, 13):
        c.cavi_step()
        e = eng.step(1, want_elbo=(it in (1, 10, 12)))
        if e is not None:
            e_cpu = c.elbo()
            # mutuality off: the ELBO carries the constant -5e5 of the nu term: absolute bound
            assert abs(e - e_cpu) <= 1e-6, (it, e, e_cpu)
    _assert_state(eng.get_state(), c, e, e_cpu)
    assert abs(eng.elbo() - e_cpu) <= 1e-6
    eng.close()


def test_config3_full_size_fit_stops_where_the_oracle_stops():
    """BASELINE configs[2] (the benchmarked configuration): the engine and the oracle, from the same start, evaluate
    the same ELBOs at iterations 1, 10, 20, ... and the reference's stop rule (model.py:1036-1056) fires at the same
    iteration; `VimureModel.fit` on the same engine reports that iteration in its trace."""
    import torch
    from oracle import cavi_coo
    from vimure_amd import CaviEngine, VimureModel
    from vimure_amd.synthetic import standard_sbm
    L, N, M, K = 4, 2000, 200, 2
    net = standard_sbm(N=N, M=M, L=L, K=K, C=2, avg_degree=5.0, sparsify=True, eta=0.5, seed=0, device="cuda:0")
    eng = CaviEngine(net.X, None, K=K, mutuality=True, device=0)
    assert eng.data_format()[0] == "sparse"
    sum_x, cov = eng.data_stats()
    seed = 1
    init = _host_state(L, N, M, K, True, seed, sum_x, cov)
    subs, vals = _coo_from_device(net.X)
    assert eng.data_format()[1] == len(vals)
    c = cavi_coo.CooRef((subs, vals), None, (L, N, N, M), K, True, PRI, *init)
    eng.set_priors(*PRI)
    eng.set_state(*init)
    coincide = [0, 0]
    old = [-1e10, -1e10]
    stop = [None, None]
    checks = []
    for it in range(1, 201):
        check = it == 1 or it % 10 == 0
        e_gpu = eng.step(1, want_elbo=check)
        c.cavi_step()
        if check:
            e_cpu = c.elbo()
            checks.append((it, e_gpu, e_cpu))
            assert abs(e_gpu - e_cpu) <= 1e-9 * abs(e_cpu), (it, e_gpu, e_cpu)
            for w, e in enumerate((e_gpu, e_cpu)):
                coincide[w] = coincide[w] + 1 if abs(e - old[w]) < 0.1 else 0
                old[w] = e
                if coincide[w] > 1 and stop[w] is None:
                    stop[w] = it
        if stop[0] is not None or stop[1] is not None:
            break
    assert stop[0] is not None and stop[0] == stop[1], (stop, checks)
    _assert_state(eng.get_state(), c, checks[-1][1], checks[-1][2], rho_rtol=1e-6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VimureModel(mutuality=True).fit(net.X, K=K, seed=seed, engine=eng, num_realisations=1, max_iter=500)
    assert int(m.trace["iter"].max()) == stop[1] and bool(m.trace["reached_convergence"].iloc[-1])
    assert abs(m.maxL - checks[-1][2]) <= 1e-9 * abs(checks[-1][2])
    eng.close()
    del net
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["default", "small_workgroups", "one_pass_two_levels", "two_passes", "no_lds_levels", "far_lists",
                                  "far_lists_two_levels", "far_lists_small_workgroups"])
def test_config5_regime_wide_reporter_dimension(mode, monkeypatch):
    """One layer in the regime of BASELINE configs[4] (M = 1000 reporters, K = 3, mutuality on): a level of the factor
    table F or of the statistics H is 24 KB here, so only a few levels of each fit in LDS beside each other.  The engine
    then either rebuilds H in a second pass (the default here and at configs[4]'s size), or keeps what fits in one pass -- the
    reports of the levels beyond added to global memory by the pass itself, or (VMR_FARL=1) kept as a compact list as well, whose
    statistics a small kernel adds after the pass; every shape is held to the oracle, including no LDS level at all."""
    import torch
    from oracle import cavi_coo
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    env = {"small_workgroups": {"VMR_TPB": "128", "VMR_ST_TPB": "256"},   # (more workgroups, fewer waves behind each ticket counter)
           "one_pass_two_levels": {"VMR_TWO_PASS": "0", "VMR_YT": "2", "VMR_HC": "2"},
           "two_passes": {"VMR_TWO_PASS": "1", "VMR_YT": "5", "VMR_HC": "5"},
           "no_lds_levels": {"VMR_TWO_PASS": "0", "VMR_YT": "0", "VMR_HC": "0", "VMR_TPB": "256"},
           "far_lists": {"VMR_FARL": "1"},
           "far_lists_two_levels": {"VMR_FARL": "1", "VMR_YT": "2", "VMR_HC": "2"},
           "far_lists_small_workgroups": {"VMR_FARL": "1", "VMR_TPB": "128", "VMR_ST_TPB": "256"}}.get(mode, {})
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    L, N, M, K = 1, 1500, 1000, 3
    net = standard_sbm(N=N, M=M, L=L, K=K, C=2, avg_degree=5.0, sparsify=True, eta=0.5, seed=2, device="cuda:0")
    eng = CaviEngine(net.X, None, K=K, mutuality=True, device=0)
    assert eng.data_format()[0] == "sparse"
    passes, levels, far = eng.sweep_shape()
    if mode.startswith("far_lists"):
        assert passes == 1 and far > 0 and (mode != "far_lists_two_levels" or levels == 2), (passes, levels, far)
    elif mode in ("default", "two_passes", "small_workgroups"):
        assert passes == 2 and far == 0, (passes, levels, far)
    sum_x, cov = eng.data_stats()
    init = _host_state(L, N, M, K, True, 3, sum_x, cov)
    subs, vals = _coo_from_device(net.X)
    c = cavi_coo.CooRef((subs, vals), None, (L, N, N, M), K, True, PRI, *init)
    eng.set_priors(*PRI)
    eng.set_state(*init)
    for it in range(1, 4):
        c.cavi_step()
        e = eng.step(1, want_elbo=True)
        e_cpu = c.elbo()
        assert abs(e - e_cpu) <= 1e-9 * abs(e_cpu), (it, e, e_cpu)
    _assert_state(eng.get_state(), c, e, e_cpu)
    assert abs(eng.elbo() - e_cpu) <= 1e-9 * abs(e_cpu)
    eng.close()
    del net
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mask", ["ones", "random"])
def test_long_steps_two_categories(mask, monkeypatch):
    """Many reports per tie (here 19 of M = 64 reporters, K = 2): steps of more rounds than the straight-line bodies hold, i.e. the
    sweep kernel's general body with its ring of loads -- against the coordinate-list oracle, mutuality on, with and without mask."""
    from oracle import cavi_coo
    from vimure_amd import CaviEngine
    monkeypatch.setenv("VMR_FORMAT", "sparse")   # (at this density the dense tiles would be chosen)
    L, N, M, K = 2, 150, 64, 2
    g = np.random.RandomState(11)
    X = ((g.rand(L, N, N, M) < 0.3) * g.randint(1, 4, size=(L, N, N, M))).astype(np.uint8)
    R = None if mask == "ones" else (g.rand(L, N, N, M) < 0.7).astype(np.uint8)
    eng = CaviEngine(X, R, K=K, mutuality=True, device=0)
    assert eng.data_format()[0] == "sparse"
    sum_x, cov = eng.data_stats()
    init = _host_state(L, N, M, K, True, 5, sum_x, cov)
    sx = np.nonzero(X)
    c = cavi_coo.CooRef((sx, X[sx]), None if R is None else np.nonzero(R), (L, N, N, M), K, True, PRI, *init)
    eng.set_priors(*PRI)
    eng.set_state(*init)
    for it in range(1, 4):
        c.cavi_step()
        e = eng.step(1, want_elbo=True)
        e_cpu = c.elbo()
        assert abs(e - e_cpu) <= 1e-9 * abs(e_cpu), (it, e, e_cpu)
    _assert_state(eng.get_state(), c, e, e_cpu)
    eng.close()


@pytest.mark.parametrize("mask", ["ones", "random"])
def test_two_passes_level_zero_rounds_with_and_without_mask(mask, monkeypatch):
    """The statistics pass of a two-pass handle with long lists: ties sorted by (reports, reports of mirror count >= 1), the
    level-0 rounds of a step add nothing to the LDS table (their marginals go through per-tie products, SlArgs::h0s) and the rho
    pass takes their factors from the per-reporter table -- K = 3, 640 reporters, two passes forced on a small network, all-ones
    and partial mask rows, against the coordinate-list oracle; VMR_NO_LEVEL0 / VMR_NO_LEVEL_SORT / VMR_NO_X0 give the same numbers."""
    from oracle import cavi_coo
    from vimure_amd import CaviEngine
    monkeypatch.setenv("VMR_FORMAT", "sparse")
    monkeypatch.setenv("VMR_TWO_PASS", "1")
    L, N, M, K = 2, 90, 640, 3
    g = np.random.RandomState(21)
    X = ((g.rand(L, N, N, M) < 0.03) * g.randint(1, 4, size=(L, N, N, M))).astype(np.uint8)
    X = np.maximum(X, (np.transpose(X, (0, 2, 1, 3)) > 0) * (g.rand(L, N, N, M) < 0.5) * g.randint(1, 3, size=(L, N, N, M))).astype(np.uint8)   # mirrored reports: levels >= 1
    R = None if mask == "ones" else (g.rand(L, N, N, M) < 0.7).astype(np.uint8)
    sx = np.nonzero(X)
    elbos = {}
    for variant in ("default", "no_level0", "no_level_sort", "no_x0"):
        if variant == "no_level0":
            monkeypatch.setenv("VMR_NO_LEVEL0", "1")
        elif variant == "no_level_sort":
            monkeypatch.delenv("VMR_NO_LEVEL0")
            monkeypatch.setenv("VMR_NO_LEVEL_SORT", "1")
        elif variant == "no_x0":   # (the level-0 rounds are walked for their counts instead of taken from the per-tie constant)
            monkeypatch.delenv("VMR_NO_LEVEL_SORT")
            monkeypatch.setenv("VMR_NO_X0", "1")
        eng = CaviEngine(X, R, K=K, mutuality=True, device=0)
        assert eng.data_format()[0] == "sparse" and eng.sweep_shape()[0] == 2
        sum_x, cov = eng.data_stats()
        init = _host_state(L, N, M, K, True, 6, sum_x, cov)
        c = cavi_coo.CooRef((sx, X[sx]), None if R is None else np.nonzero(R), (L, N, N, M), K, True, PRI, *init)
        eng.set_priors(*PRI)
        eng.set_state(*init)
        for it in range(1, 4):
            c.cavi_step()
            e = eng.step(1, want_elbo=True)
            e_cpu = c.elbo()
            assert abs(e - e_cpu) <= 1e-9 * abs(e_cpu), (variant, it, e, e_cpu)
        _assert_state(eng.get_state(), c, e, e_cpu)
        elbos[variant] = e
        eng.close()
    assert max(elbos.values()) - min(elbos.values()) <= 1e-9 * abs(elbos["default"])


def test_config5_layer_at_stated_size():
    """BASELINE configs[4] as stated -- L=8, N=8000, M=1000, K=3, one layer per GPU: ONE GPU's share (a 64 GB layer generated on
    the device, 1.2 G reports) through a sweep, checked by properties that need no 200-second oracle build:
      * every tie's rho sums to 1 or -- where the reference's raw exponentials all underflow, which with 1000 reporters' worth
        of E[theta] in the exponent happens at the first sweeps -- is left all zero (model.py:807-811);
      * rho of 20 000 sampled ties recomputed on the host from the engine's own gamma / phi / nu and those ties' reports -- the
        rho update is per tie given the parameter tables (model.py:795-811, 889-923);
      * gamma_rte from the all-ones mask sums of the prior (model.py:704-718);
      * the mass identity sum_m (gamma_shp - alpha) + (nu partial) = sum over reports of x sum_k rho_k at fixed parameters and
        rho: w1 + w2 = 1 per report (model.py:685-696), i.e. every one of the layer's reports was counted exactly once;
      * the ELBO fused into the sweep equals the stand-alone one (model.py:948-1019).
    The N=1500 oracle test above stays the bit-level check of this regime."""
    import scipy.special as sp
    import torch
    from vimure_amd import CaviEngine, _lib
    from vimure_amd.synthetic import standard_sbm
    L, N, M, K = 1, 8000, 1000, 3
    EPS = 1e-12
    net = standard_sbm(N=N, M=M, L=L, K=K, C=2, avg_degree=5.0, sparsify=True, eta=0.5, seed=0, device="cuda:0")
    eng = CaviEngine(net.X, None, K=K, mutuality=True, device=0)
    assert eng.data_format()[0] == "sparse"
    sum_x, cov = eng.data_stats()
    init = _host_state(L, N, M, K, True, 1, sum_x, cov)
    g_shp0, g_rte0, p_shp0, p_rte0, nu_shp0, nu_rte0, pr = init
    eng.set_priors(*PRI)
    eng.set_state(*init)
    nu_part, e_main, e_q = eng.sweep_local(want_elbo=True)   # gamma, phi, rho (+ ELBO terms); nu NOT committed
    st = eng.get_state(rho=True)
    rho = st["rho"]
    assert np.isfinite(rho).all()
    rsum = rho.sum(axis=-1)
    zero_rows = (rho == 0.0).all(axis=-1)   # every exp(a_k) underflowed: left unnormalised, as the reference does
    assert np.abs(rsum[~zero_rows] - 1.0).max() < 1e-12
    assert st["nu_shp"] == nu_shp0   # (not committed)
    # gamma_rte = beta + sum_k E[lambda_k]_old sum_t pr_rho_k: the mask is all ones (the pass that sums rho over ties)
    S0 = pr.reshape(-1, K).sum(axis=0)
    np.testing.assert_allclose(st["gamma_rte"][0], PRI[1] + float(((p_shp0 / p_rte0)[0] * S0).sum()), rtol=1e-11)
    # sampled ties
    g = np.random.RandomState(5)
    n = 20000
    ii, jj = g.randint(0, N, n), g.randint(0, N, n)
    di, dj = torch.as_tensor(ii, device="cuda:0"), torch.as_tensor(jj, device="cuda:0")
    xs = net.X[0, di, dj].cpu().numpy().astype(np.float64)    # [n, M] the ties' reports
    ys = net.X[0, dj, di].cpu().numpy().astype(np.float64)    # mirrored counts
    lth = sp.psi(st["gamma_shp"][0]) - np.log(st["gamma_rte"][0])
    lla = sp.psi(st["phi_shp"][0]) - np.log(st["phi_rte"][0])
    Eth, Ela = st["gamma_shp"][0] / st["gamma_rte"][0], st["phi_shp"][0] / st["phi_rte"][0]
    gnu = float(np.exp(sp.psi(st["nu_shp"]) - np.log(st["nu_rte"])))
    z2 = gnu * ys                                              # [n, M]
    a = np.log(pr[0, ii, jj] + EPS) - Eth.sum() * Ela[None, :]
    for k in range(K):
        z1 = (np.exp(lth) * np.exp(lla[k]))[None, :]
        den = z1 + z2
        den[den == 0.0] = 1.0
        a[:, k] += (xs * (lth[None, :] + lla[k]) * (z1 / den)).sum(axis=1)
    ex = np.exp(a)
    sm = ex.sum(axis=1, keepdims=True)
    want = np.where(sm > 0, ex / np.where(sm > 0, sm, 1.0), ex)
    np.testing.assert_allclose(rho[0, ii, jj], want, rtol=1e-9, atol=1e-13)
    # mass identity at fixed parameters and rho: the gamma sub-step now uses the same theta, lambda, nu and rho as nu_part did
    eng.sub_step(_lib.STEP_GAMMA)
    g2 = eng.get_state(rho=False)["gamma_shp"][0]
    xt = torch.cat([net.X[0, i0:i0 + 200].sum(dim=-1, dtype=torch.int64) for i0 in range(0, N, 200)]).cpu().numpy()   # reports per tie
    mass = float((xt * rsum[0]).sum())
    assert abs((g2 - PRI[0]).sum() + nu_part - mass) <= 1e-10 * sum_x, ((g2 - PRI[0]).sum(), nu_part, mass, sum_x)
    # a fused sweep + ELBO at full size, against the stand-alone ELBO pass
    eng.set_state(*init)
    e1 = eng.step(2, want_elbo=True)
    assert np.isfinite(e1)
    assert abs(eng.elbo() - e1) <= 1e-9 * abs(e1)
    eng.close()
    del net
    torch.cuda.empty_cache()
