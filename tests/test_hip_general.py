"""GPU: inputs beyond the specialised kernels -- more than 8 categories (the reference's DEFAULT is K = max(X) + 1, model.py:179-197,
exercised by its test/test_model.py:59-115) and counts beyond the 11 bits / 2^20 table rows of a packed entry (the reference holds
int64 counts, utils.py:241-242).  The general kernels (csrc/sweep_gen.hip) against the coordinate-list oracle, the NumPy oracle
and golden vectors of the reference (L_default_K12, M_K16_nomut, O_wide_rows, N_counts_12000; the first three also run through
tests/test_hip_substeps.py, test_hip_coo.py and test_hip_fit.py with every other golden case)."""
import os
import warnings

import numpy as np
import pytest

from oracle import cavi_coo
from oracle import vimure_oracle as vo

pytestmark = pytest.mark.gpu
PRI = (0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _network(L, N, M, K, xmax, seed, mask, dens=0.08):
    g = np.random.RandomState(seed)
    X = ((g.rand(L, N, N, M) < dens) * g.randint(1, xmax + 1, size=(L, N, N, M))).astype(np.int64)
    if mask == "ones":
        R = None
    elif mask == "words":      # long partial rows: bit-packed mask words
        R = (g.rand(L, N, N, M) < 0.9).astype(np.uint8)
        R[:, :2] = 1           # some all-ones rows
        R[:, 2] = 0            # some empty ones
    else:                      # "lists": the self-reporter mask of survey data (short partial rows)
        R = np.zeros((L, N, N, M), np.uint8)
        for m in range(min(M, N)):
            R[:, m, :, m] = 1
            R[:, :, m, m] = 1
    return X, R


def _pair(X, R, K, mut, seed, coo=True, gamma=None, phi=None):
    """(engine, oracle) from the same seeded state."""
    from vimure_amd import CaviEngine
    L, N, _, M = X.shape
    g = np.random.RandomState(seed)
    pr = g.rand(L, N, N, K) + 0.05
    pr /= pr.sum(-1, keepdims=True)
    gs, gr = gamma if gamma is not None else (0.5 + g.rand(L, M), 0.5 + g.rand(L, M))
    ps, prt = phi if phi is not None else (1 + g.rand(L, K), 1 + g.rand(L, K))
    init = (gs, gr, ps, prt, 0.7 if mut else 1e-6, 1.0 + float(X.sum()) if mut else 1.0, pr)
    sx = np.nonzero(X)
    Rs = None if R is None else np.nonzero(R)
    if coo:
        eng = CaviEngine.from_coo(sx, X[sx], X.shape, R=Rs, K=K, mutuality=mut)
    else:
        eng = CaviEngine(X.astype(np.uint8), R, K=K, mutuality=mut)
    c = cavi_coo.CooRef((sx, X[sx].astype(np.int32)), Rs, X.shape, K, mut, PRI, *init)
    eng.set_priors(*PRI)
    eng.set_state(*init)
    return eng, c


def _same_state(eng, c, rtol=1e-9):
    st = eng.get_state()
    for n in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte"):
        np.testing.assert_allclose(st[n], getattr(c, n), rtol=rtol, err_msg=n)
    np.testing.assert_allclose(st["rho"], c.rho, rtol=1e-8, atol=1e-13)
    assert abs(st["nu_shp"] - c.nu_shp) <= rtol * abs(c.nu_shp)


@pytest.mark.parametrize("K", [9, 16, 21])
@pytest.mark.parametrize("mask", ["ones", "words", "lists"])
def test_more_than_eight_categories_substeps_and_sweeps(K, mask):
    """K in {9, 16, 21}: every sub-step from a seeded state, then fused sweeps with the ELBO, against the coordinate-list oracle;
    through the coordinate-list entry point and (all-ones / word masks) the dense one."""
    from vimure_amd import _lib
    L, N, M = 2, 23, {"lists": 23, "words": 80, "ones": 11}[mask]   # (rows of more than 64 reporters stay mask words)
    X, R = _network(L, N, M, K, 6, seed=K, mask=mask)
    for coo in ([True, False] if mask != "lists" else [True]):
        eng, c = _pair(X, R, K, True, seed=3, coo=coo)
        assert eng.data_format()[0] == "sparse"
        assert eng.mask_format()[0] == ("lists" if mask == "lists" else "words")
        eng.sub_step(_lib.STEP_GAMMA); c.update_gamma()
        g = eng.get_state(rho=False)
        np.testing.assert_allclose(g["gamma_shp"], c.gamma_shp, rtol=1e-9)
        np.testing.assert_allclose(g["gamma_rte"], c.gamma_rte, rtol=1e-9)
        eng.sub_step(_lib.STEP_PHI); c.update_phi()
        g = eng.get_state(rho=False)
        np.testing.assert_allclose(g["phi_shp"], c.phi_shp, rtol=1e-9)
        np.testing.assert_allclose(g["phi_rte"], c.phi_rte, rtol=1e-9)
        eng.sub_step(_lib.STEP_RHO); c.update_rho()
        np.testing.assert_allclose(eng.get_state()["rho"], c.rho, rtol=1e-9, atol=1e-13)
        eng.sub_step(_lib.STEP_NU); c.update_nu()
        assert abs(eng.get_state(rho=False)["nu_shp"] - c.nu_shp) <= 1e-9 * abs(c.nu_shp)
        e, ec = eng.elbo(), c.elbo()
        assert abs(e - ec) <= 1e-10 * max(1.0, abs(ec)), (e, ec)
        for _ in range(3):
            c.cavi_step()
            e, ec = eng.step(1, want_elbo=True), c.elbo()
            assert abs(e - ec) <= 1e-10 * max(1.0, abs(ec)), (e, ec)
        _same_state(eng, c)
        eng.close()


@pytest.mark.parametrize("K", [70, 130])
def test_categories_beyond_one_wave(K):
    """K > 64: several categories per lane (NCH = 2 and 4 in k_sweep_gen)."""
    X, R = _network(1, 12, 70, K, 4, seed=K, mask="words", dens=0.05)
    eng, c = _pair(X, R, K, True, seed=5)
    for _ in range(3):
        c.cavi_step()
        e, ec = eng.step(1, want_elbo=True), c.elbo()
        assert abs(e - ec) <= 1e-10 * max(1.0, abs(ec)), (e, ec)
    _same_state(eng, c)
    eng.close()


@pytest.mark.parametrize("K,mut", [(9, True), (21, True), (70, True), (16, False)])
def test_sums_over_the_statistics_by_several_workgroups(K, mut, monkeypatch):
    """Large statistics tables are summed by several workgroups per layer (k_gen_hsum) before / after the finalize kernel: forced
    here on small ones (VMR_GEN_HSUM), sub-steps and fused sweeps against the oracle as above."""
    from vimure_amd import _lib
    monkeypatch.setenv("VMR_GEN_HSUM", "3")
    X, R = _network(2, 19, 70, K, 5, seed=K + 1, mask="words", dens=0.05)
    eng, c = _pair(X, R, K, mut, seed=4)
    eng.sub_step(_lib.STEP_GAMMA); c.update_gamma()
    g = eng.get_state(rho=False)
    np.testing.assert_allclose(g["gamma_shp"], c.gamma_shp, rtol=1e-9)
    np.testing.assert_allclose(g["gamma_rte"], c.gamma_rte, rtol=1e-9)
    eng.sub_step(_lib.STEP_PHI); c.update_phi()
    g = eng.get_state(rho=False)
    np.testing.assert_allclose(g["phi_shp"], c.phi_shp, rtol=1e-9)
    np.testing.assert_allclose(g["phi_rte"], c.phi_rte, rtol=1e-9)
    for _ in range(3):
        c.cavi_step()
        eng.step(1)
    st = eng.get_state()
    for n in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte"):
        np.testing.assert_allclose(st[n], getattr(c, n), rtol=1e-9, err_msg=n)
    np.testing.assert_allclose(st["rho"], c.rho, rtol=1e-8, atol=1e-13)
    eng.close()


def test_mutuality_off_sixteen_categories():
    X, R = _network(2, 20, 72, 16, 5, seed=8, mask="words", dens=0.03)
    eng, c = _pair(X, R, 16, False, seed=2)
    for _ in range(3):
        c.cavi_step()
        e, ec = eng.step(1, want_elbo=True), c.elbo()
        assert abs(e - ec) <= 1e-6 * max(1.0, abs(ec + 5e5)) + 1e-9 * abs(ec), (e, ec)   # (the -5e5 offset of SURVEY App. C 14)
    st = eng.get_state()
    np.testing.assert_allclose(st["rho"], c.rho, rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(st["phi_shp"], c.phi_shp, rtol=1e-9)
    eng.close()


@pytest.mark.parametrize("mask", ["ones", "lists"])
def test_counts_to_twelve_thousand(mask):
    """Counts far beyond a packed entry's 11 bits (two-word entries).  The state is chosen so that the reference's raw exponentials
    stay finite (E log theta + E log lambda slightly negative: x times it is a few tens) -- from its own initial state the
    reference overflows on such counts (test_reference_nans_on_huge_counts_and_so_do_we) -- and every sub-step is taken from it."""
    from vimure_amd import _lib
    L, N, M, K = 1, 21, 21, 2
    X, R = _network(L, N, M, K, 12000, seed=4, mask=mask, dens=0.05)
    X[0, 1, 2, 3] = 12000
    X[0, 2, 1, 3] = 11999
    if R is not None:
        R[0, 1, 2, 3] = R[0, 2, 1, 3] = 1
    big = (np.full((L, M), 4000.0), np.full((L, M), 4002.0))   # E log theta = psi(4000) - log(4002) = -6.2e-4
    lam = (np.full((L, K), 50.0), np.array([[51.0, 52.0]]))    # E log lambda = psi(50) - log(51 | 52) = -0.03 | -0.05
    for step, upd in ((_lib.STEP_GAMMA, "update_gamma"), (_lib.STEP_PHI, "update_phi"), (_lib.STEP_RHO, "update_rho"), (_lib.STEP_NU, "update_nu")):
        eng, c = _pair(X, R, K, True, seed=6, gamma=big, phi=lam)
        assert eng.data_stats()[0] == float(X.sum())
        if step == _lib.STEP_PHI:   # (the PHI commit follows a GAMMA sub-step, as in a sweep)
            eng.sub_step(_lib.STEP_GAMMA); c.update_gamma()
        if step == _lib.STEP_NU:
            eng.sub_step(_lib.STEP_RHO); c.update_rho()
        eng.sub_step(step); getattr(c, upd)()
        _same_state(eng, c)
        assert np.isfinite(c.rho).all()
        if step == _lib.STEP_RHO:
            e, ec = eng.elbo(), c.elbo()
            assert np.isfinite(ec) and abs(e - ec) <= 1e-10 * max(1.0, abs(ec)), (e, ec)
        eng.close()


def test_reference_nans_on_huge_counts_and_so_do_we():
    """Golden N_counts_12000: the reference accepts the tensor (counts to 16990) and then raises "ELBO is NaN!!!!" at its first
    ELBO check (model.py:1016) -- its raw exponentials overflow.  Same inputs, same error here: the engine takes the counts
    (two-word entries) and the fit raises the reference's message."""
    from vimure_amd import VimureModel
    from vimure_amd.tensor import SparseTensor
    d = dict(np.load(os.path.join(GOLDEN, "N_counts_12000.npz")))
    assert str(d["error"]) == "ELBO is NaN!!!!"
    shape = tuple(int(v) for v in d["X_shape"])
    X = SparseTensor(tuple(d["X_subs"].astype(np.int64)), d["X_vals"].astype(np.int64), shape=shape)
    R = SparseTensor(tuple(d["R_subs"].astype(np.int64)), np.ones(d["R_subs"].shape[1], np.int64), shape=shape)
    for Xin in (X, X.toarray()):   # the container, and the dense int64 array the reference was given
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with pytest.raises(ValueError, match="ELBO is NaN!!!!"):
                VimureModel().fit(Xin, R=R, K=int(d["K"]), seed=int(d["seed"]), num_realisations=1, max_iter=20)


def test_fit_with_default_k_on_sbm_counts():
    """`fit(X)` without K on a StandardSBM(K=3) tensor: K = max(X) + 1 (here > 8), the whole fit against the NumPy oracle."""
    from vimure_amd import VimureModel
    from vimure_amd.synthetic import standard_sbm
    net = standard_sbm(N=40, M=12, L=1, K=3, avg_degree=6.0, eta=0.3, seed=11)
    X = net.X.astype(np.int64)
    X[X > 0] += np.random.RandomState(1).randint(0, 7, size=int((X > 0).sum()))   # counts to ~10
    Kd = int(X.max()) + 1
    assert Kd > 8
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = VimureModel().fit(X.astype(np.uint8), seed=3, num_realisations=1, max_iter=30)
    assert any("Defaulting to" in str(x.message) for x in w) and m.K == Kd
    pr = vo.make_priors(1, 12, Kd)
    pb = vo.Problem(X.astype(np.uint8), np.ones(X.shape, np.uint8), Kd, True, pr)
    res = vo.fit(pb, seed=3, num_realisations=1, max_iter=30)
    assert m.trace["iter"].tolist() == [t[2] for t in res.trace]
    np.testing.assert_allclose(m.trace["elbo"].values, [t[3] for t in res.trace], rtol=1e-9)
    assert abs(m.maxL - res.maxL) <= 1e-9 * abs(res.maxL)
    np.testing.assert_allclose(m.rho_f, res.best.rho, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(m.gamma_shp_f, res.best.gamma_shp, rtol=1e-8)


def test_readout_and_sampler_with_twelve_categories():
    from vimure_amd import CaviEngine
    L, N, M, K = 1, 30, 5, 12
    X, R = _network(L, N, M, K, 5, seed=2, mask="ones")
    eng, c = _pair(X, R, K, True, seed=9)
    eng.step(2)
    rho = eng.get_state()["rho"]
    assert np.array_equal(eng.readout("rho_max"), rho.argmax(-1).astype(np.uint8))
    np.testing.assert_allclose(eng.readout("rho_mean"), (rho * np.arange(K)).sum(-1), rtol=1e-12)
    # the sampler: per tie the most frequent of n categorical trials; with many trials that is the arg max wherever it is clear
    y = eng.sample(seed=5, n_trials=4000)
    top2 = np.sort(rho, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 0.08   # (4000 trials: a margin of 320 counts against a deviation of ~35)
    assert clear.sum() > 5 and np.array_equal(y[clear], rho.argmax(-1)[clear].astype(np.uint8))
    one = eng.sample(seed=6, n_trials=1)    # single draws: the category frequencies follow rho
    freq = np.bincount(one.ravel(), minlength=K) / one.size
    np.testing.assert_allclose(freq, rho.reshape(-1, K).mean(0), atol=4.0 / np.sqrt(one.size))
    eng.close()


def test_lockstep_batch_runs_general_handles_on_their_own():
    """vmr_fit_loop_batch with handles the lockstep launch does not serve (K > 8): each runs its own loop, same results."""
    from vimure_amd import CaviEngine
    outs = []
    for batch in (False, True):
        engs, refs = [], []
        for s, K in ((1, 12), (2, 2), (3, 12)):
            X, R = _network(1, 18, 18, K, 4, seed=s, mask="lists")
            eng, c = _pair(X, R, K, True, seed=s)
            engs.append(eng)
        if batch:
            res = CaviEngine.fit_loop_batch(engs, 30, 0.1, 1)
        else:
            res = [e.fit_loop(30, 0.1, 1) for e in engs]
        outs.append([(r[1], r[2], r[3]) for r in res])
        for e in engs:
            e.close()
    for a, b in zip(*outs):
        assert a[1:] == b[1:] and abs(a[0] - b[0]) <= 1e-9 * abs(a[0])


def test_layer_sharded_sweeps_of_a_general_handle():
    """vmr_sweep_local / vmr_commit_nu (the layer-sharded exchange) on a handle of the general kernels: with one owner the exchange is
    the identity, so it must leave what vmr_step leaves; and the raw pieces must assemble to the oracle's ELBO."""
    import scipy.special as sp
    K = 12
    X, R = _network(2, 20, 9, K, 6, seed=3, mask="ones")
    a, c = _pair(X, R, K, True, seed=4)
    b, _ = _pair(X, R, K, True, seed=4)
    for it in range(3):
        a.step(1)
        c.cavi_step()
        nu_part, e_main, e_q = b.sweep_local(want_elbo=(it == 2))
        b.commit_nu(nu_part)
    sa, sb = a.get_state(), b.get_state()
    for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "rho"):
        np.testing.assert_allclose(sb[k], sa[k], rtol=1e-10, atol=1e-13, err_msg=k)
    nu_shp, nu_rte = sb["nu_shp"], sb["nu_rte"]
    elbo = e_main - (nu_shp / nu_rte) * e_q + float(sp.gammaln(nu_shp) - PRI[4] * np.log(nu_rte) + (PRI[4] - nu_shp) * sp.psi(nu_shp)
                                                   + nu_shp * (1.0 - PRI[5] / nu_rte))
    ec = c.elbo()
    assert abs(elbo - ec) <= 1e-9 * max(1.0, abs(ec)), (elbo, ec)
    a.close(); b.close()
