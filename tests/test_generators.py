"""The generator classes (vimure_amd.synthetic: StandardSBM, DegreeCorrectedSBM, Multitensor, `_build_X`, baselines,
build_custom_theta, build_self_reporter_mask) against what the REAL reference produces for the same arguments and
seeds (tests/golden/K_generators.npz, dumped by tools/make_golden_generators.py): exact mode follows the reference's
RandomState stream bit for bit.  The whole-array mode draws from the same distributions: checked statistically."""
import os
import warnings

import numpy as np
import pytest

from tests.golden_util import GOLDEN
from vimure_amd import synthetic as sy

G = dict(np.load(os.path.join(GOLDEN, "K_generators.npz")))
CLS = {"sbm": sy.StandardSBM, "sbm_nosparse": sy.StandardSBM, "dcsbm": sy.DegreeCorrectedSBM, "multitensor": sy.Multitensor}


def _args(name, prefix):
    out = {}
    for k, v in G.items():
        if k.startswith(f"{name}_{prefix}_"):
            key = k[len(f"{name}_{prefix}_"):]
            out[key] = v.item() if v.ndim == 0 else v
    return out


def _make(name):
    kw = _args(name, "arg")
    if f"{name}_structure" in G:
        kw["structure"] = [str(s) for s in G[f"{name}_structure"]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return CLS[name](**kw)


@pytest.mark.parametrize("name", ["sbm", "sbm_nosparse", "dcsbm", "multitensor"])
def test_ground_truth_is_the_references(name):
    net = _make(name)
    assert np.array_equal(net.Y.toarray(), G[f"{name}_Y"])
    assert net.Y.shape == (net.L, net.N, net.N) and hasattr(net.Y, "subs") and hasattr(net.Y, "vals")
    np.testing.assert_array_equal(net.u, G[f"{name}_u"])
    np.testing.assert_array_equal(net.v, G[f"{name}_v"])
    np.testing.assert_allclose(net.w, G[f"{name}_w"], rtol=1e-15)
    if name == "dcsbm":
        assert np.array_equal(net.d_in, G["dcsbm_d_in"]) and np.array_equal(net.d_out, G["dcsbm_d_out"])


@pytest.mark.parametrize("name", ["sbm", "sbm_nosparse", "dcsbm", "multitensor"])
def test_build_X_is_the_references(name):
    net = _make(name)
    b = _args(name, "build")
    if name == "multitensor":
        b["theta"] = sy.build_custom_theta(net, theta_ratio=0.1, exaggeration_type="over", seed=25)
        assert np.array_equal(b["theta"], G["multitensor_custom_theta"])
    net._build_X(exact=True, **b)
    assert np.array_equal(net.X.toarray(), G[f"{name}_X"])
    assert np.array_equal(net.R.toarray(), G[f"{name}_R"])
    np.testing.assert_array_equal(net.theta, G[f"{name}_theta"])
    np.testing.assert_array_equal(net.lambda_k, G[f"{name}_lambda_k"])
    assert np.array_equal(net.X_union.toarray(), G[f"{name}_X_union"])
    assert (net.X_intersection is not None) == bool(G[f"{name}_has_intersection"])
    if net.X_intersection is not None:
        assert np.array_equal(net.X_intersection.toarray(), G[f"{name}_X_intersection"])


def test_mask_and_custom_theta():
    net = sy.StandardSBM(N=12, M=12, L=2, K=2, seed=0)
    assert np.array_equal(sy.build_self_reporter_mask(net), G["mask_self"])
    assert np.array_equal(sy.build_custom_theta(net, theta_ratio=0.5, exaggeration_type="under", seed=4), G["custom_theta_under"])
    assert np.array_equal(sy.build_custom_theta(net, theta_ratio=0.25, exaggeration_type="over", seed=9), G["custom_theta_over"])
    with pytest.raises(ValueError, match="theta_ratio"):
        sy.build_custom_theta(net, theta_ratio=1.5)
    with pytest.raises(ValueError, match="exaggeration_type"):
        sy.build_custom_theta(net, exaggeration_type="sideways")
    with pytest.raises(ValueError, match="mutuality"):
        net._build_X(mutuality=1.0)
    with pytest.raises(ValueError, match="structures"):
        sy.StandardSBM(N=10, M=10, structure="random")


def _moments(X, R):
    X = X.toarray().astype(float)
    R = R.toarray()
    on = R > 0
    recip = (X * X.transpose(0, 2, 1, 3))[on].mean()
    return X[on].mean(), (X[on] > 0).mean(), recip


def test_whole_array_mode_draws_from_the_same_distributions():
    """exact=False is another stream of the same model: means, densities and reciprocity agree with exact mode over seeds."""
    net = sy.StandardSBM(N=60, M=60, L=1, K=2, C=2, avg_degree=6, seed=3)
    th = np.random.RandomState(0).gamma(2.0, 0.5, size=(1, 60))   # the same reliabilities for both modes
    a, b = [], []
    for s in range(6):
        a.append(_moments(*(lambda n: (n.X, n.R))(net._build_X(mutuality=0.4, seed=s, theta=th, exact=True))))
        b.append(_moments(*(lambda n: (n.X, n.R))(net._build_X(mutuality=0.4, seed=100 + s, theta=th, exact=False))))
    a, b = np.mean(a, axis=0), np.mean(b, axis=0)
    assert np.all(np.abs(a - b) <= 0.2 * np.abs(a) + 1e-3), (a, b)
    # the all-ones branch visits every pair once in both modes
    net2 = sy.StandardSBM(N=40, M=8, L=1, K=2, C=2, avg_degree=6, seed=1)
    th2 = np.random.RandomState(1).gamma(2.0, 0.5, size=(1, 8))
    e = _moments(*(lambda n: (n.X, n.R))(net2._build_X(mutuality=0.3, flag_self_reporter=False, seed=1, theta=th2, exact=True)))
    v = _moments(*(lambda n: (n.X, n.R))(net2._build_X(mutuality=0.3, flag_self_reporter=False, seed=2, theta=th2, exact=False)))
    assert np.all(np.abs(np.array(e) - np.array(v)) <= 0.25 * np.abs(np.array(e)) + 1e-3), (e, v)
    # Multitensor: the pair-by-pair stream of the reference against whole-array pairs (its default above N = 600)
    ex = [sy.Multitensor(N=150, M=150, K=2, avg_degree=8, eta=0.3, seed=s, exact=True).Y.toarray()[0] for s in range(3)]
    va = [sy.Multitensor(N=150, M=150, K=2, avg_degree=8, eta=0.3, seed=10 + s, exact=False).Y.toarray()[0] for s in range(3)]
    dens = lambda Ys: np.mean([Y.sum() for Y in Ys])
    rec = lambda Ys: np.mean([(Y * Y.T).sum() / Y.sum() for Y in Ys])
    assert abs(dens(ex) - dens(va)) <= 0.1 * dens(ex) and abs(rec(ex) - rec(va)) <= 0.25 * rec(ex), (dens(ex), dens(va), rec(ex), rec(va))
    assert all(np.all(np.diag(Y) == 0) for Y in va) and rec(va) > 3 * dens(va) / 150 ** 2   # reciprocity far above chance


@pytest.mark.gpu
def test_whole_array_mode_on_the_gpu():
    net = sy.StandardSBM(N=80, M=80, L=2, K=2, C=2, avg_degree=6, seed=3)
    th = np.random.RandomState(0).gamma(2.0, 0.5, size=(2, 80))
    cpu = _moments(*(lambda n: (n.X, n.R))(net._build_X(mutuality=0.4, seed=1, theta=th, exact=False)))
    gpu = _moments(*(lambda n: (n.X, n.R))(net._build_X(mutuality=0.4, seed=2, theta=th, exact=False, device="cuda:0")))
    assert np.all(np.abs(np.array(cpu) - np.array(gpu)) <= 0.2 * np.abs(np.array(cpu)) + 1e-3), (cpu, gpu)
    assert net.X.toarray().shape == (2, 80, 80, 80) and np.all(net.X.toarray()[net.R.toarray() == 0] == 0)
    from vimure_amd.synthetic import standard_sbm
    d = standard_sbm(N=120, M=30, L=2, K=2, avg_degree=5.0, eta=0.4, seed=0, device="cuda:0")   # the blocked large-scale generator
    h = standard_sbm(N=120, M=30, L=2, K=2, avg_degree=5.0, eta=0.4, seed=0)
    xd, xh = d.X.cpu().numpy().astype(float), h.X.astype(float)
    assert abs(xd.mean() - xh.mean()) <= 0.15 * xh.mean()
    assert abs((xd * xd.transpose(0, 2, 1, 3)).mean() - (xh * xh.transpose(0, 2, 1, 3)).mean()) <= 0.3 * (xh * xh.transpose(0, 2, 1, 3)).mean()


def test_posterior_predictive_network_is_the_references():
    """`PosteriorSyntheticNetwork` (reference synthetic.py:964-1177) on the same "fitted model": Y, the Gamma draws and X."""
    class Fitted:
        pass
    fm = Fitted()
    for k in ("rho_f", "gamma_shp_f", "gamma_rte_f", "phi_shp_f", "phi_rte_f", "nu_shp_f", "nu_rte_f"):
        setattr(fm, k, G["post_" + k])
    pn = sy.PosteriorSyntheticNetwork(fm, seed_Y=7)
    pn.build_Y()
    assert np.array_equal(pn.Y.toarray(), G["post_Y"])
    pn.build_X(flag_self_reporter=True, seed_X=5, exact=True)
    np.testing.assert_array_equal(pn.theta, G["post_theta"])
    np.testing.assert_array_equal(pn.lambda_k, G["post_lambda_k"])
    assert pn.mutuality == float(G["post_mutuality"])
    np.testing.assert_array_equal(pn.lambda_k_auxiliary, G["post_lambda_aux"])
    assert np.array_equal(pn.X.toarray(), G["post_X"]) and np.array_equal(pn.R.toarray(), G["post_R"])
    pn.build_X(flag_self_reporter=False, cutoff_X=True, seed_X=6, exact=True)
    assert np.array_equal(pn.X.toarray(), G["post_X_ones"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sbm", "sbm_nosparse", "dcsbm"])
def test_device_generator_kernel_against_the_references_draws_by_moments(name):
    """The HIP generator (csrc/generate.hip, vmr_generate_x) draws `_build_X` (reference synthetic.py:159-231) from a Philox stream,
    so it is held to the REFERENCE's own output (fixture K, dumped from the real `_build_X`) statistically: over 300 seeds, per
    (layer, reporter): the mean count, the density of non-zero counts and the reciprocity sum_ij X_ij X_ji of the reference's draw
    must lie inside the kernel's distribution (inside the range of the 300 draws cell by cell, |z| < 4.5 pooled over a layer's
    reporters), and the ensemble mean of every cell must match the analytic mean (lambda theta + eta mirror) / (1 - eta^2) within 5 standard errors."""
    import torch
    theta, lam = G[f"{name}_theta"], G[f"{name}_lambda_k"]
    eta, self_rep = float(G[f"{name}_build_mutuality"]), bool(G[f"{name}_build_flag_self_reporter"])
    Xref = G[f"{name}_X"].astype(np.float64)
    if f"{name}_build_cutoff_X" in G and bool(G[f"{name}_build_cutoff_X"]):
        pytest.skip("the fixture's draw was cut off at Q - 1 (a host-side step after the draw)")
    L, N, _, M = Xref.shape
    lam_d = torch.as_tensor(lam, device="cuda:0", dtype=torch.float64)
    n = 300
    acc = np.zeros(Xref.shape)
    stats = np.zeros((n, 3, L, M))

    def st(X):
        return np.stack([X.mean(axis=(1, 2)), (X > 0).mean(axis=(1, 2)), (X * X.transpose(0, 2, 1, 3)).sum(axis=(1, 2))])
    for s in range(n):
        X = sy.device_build_x(None, theta, eta, 1000 + s, lam=lam_d, flag_self_reporter=self_rep).cpu().numpy().astype(np.float64)
        assert np.all(X[:, np.arange(N), np.arange(N)] == 0)
        if self_rep:
            R = sy.self_reporter_mask(L, N, M)
            assert np.all(X[R == 0] == 0)
        acc += X
        stats[s] = st(X)
    ref = st(Xref)
    # per (statistic, layer, reporter): counts this small are far from normal -- the reference's value must lie inside the range the
    # 300 draws span, up to the few exceedances 3 L M such comparisons produce by chance (2 / 301 each)
    outside = int(((ref < stats.min(0)) | (ref > stats.max(0))).sum())
    assert outside <= 4 + 3 * ref.size // 100, (outside, ref.size)
    # pooled over the reporters of a layer the statistics are sums of hundreds of terms: a z-test against the ensemble's spread
    pooled, pref = stats.sum(axis=3), ref.sum(axis=2)
    z = (pref - pooled.mean(0)) / (pooled.std(0) + 1e-12)
    assert np.all(np.abs(z) < 4.5), z
    MX = theta[:, None, None, :] * lam[..., None]
    MM = (MX + eta * MX.transpose(0, 2, 1, 3)) / (1.0 - eta * eta)
    if self_rep:
        MM = MM * sy.self_reporter_mask(L, N, M)
    MM[:, np.arange(N), np.arange(N)] = 0.0
    var = MM * (1.0 + 0.5 * eta * eta * 2.0) + 1e-12      # (a mixture of a Poisson and a Poisson given a Poisson: a little over-dispersed)
    err = np.abs(acc / n - MM) / np.sqrt(var / n)
    assert err.max() < 6.0, float(err.max())
    assert abs(acc.sum() / n - MM.sum()) <= 5.0 * np.sqrt(var.sum() / n)


@pytest.mark.gpu
def test_device_generator_is_a_function_of_the_seed_and_fills_the_engine():
    """Same seed, same tensor (whatever the launch); the Y kernel's block structure; large rates (PTRS branch) by their mean."""
    import torch
    from vimure_amd.synthetic import standard_sbm
    a = standard_sbm(N=150, M=40, L=2, K=3, avg_degree=6.0, eta=0.4, seed=5, device="cuda:0")
    b = standard_sbm(N=150, M=40, L=2, K=3, avg_degree=6.0, eta=0.4, seed=5, device="cuda:0")
    c = standard_sbm(N=150, M=40, L=2, K=3, avg_degree=6.0, eta=0.4, seed=6, device="cuda:0")
    assert torch.equal(a.X, b.X) and torch.equal(a.Y, b.Y) and not torch.equal(a.X, c.X)
    Y = a.Y.cpu().numpy()
    assert Y.max() <= 2 and np.all(Y[:, np.arange(150), np.arange(150)] == 0)
    grp = np.minimum(np.arange(150) // 75, 1)
    same = grp[:, None] == grp[None, :]
    assert Y[:, same].mean() > 4 * Y[:, ~same].mean()          # assortative: ten times the ties inside a group
    big = sy.device_build_x(None, np.full((1, 64), 3.0), 0.2, 7, lam=torch.full((1, 40, 40), 40.0, device="cuda:0", dtype=torch.float64))
    m = big.cpu().numpy().astype(float)
    off = ~np.eye(40, dtype=bool)
    expect = (120.0 + 0.2 * 120.0) / (1 - 0.04)
    assert abs(m[0][off].mean() - expect) < 0.02 * expect and m.max() <= 255
