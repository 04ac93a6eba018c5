"""Host logic of vimure_amd.VimureModel that runs before the engine: argument validation with the
reference's messages (model.py:79-325), RandomState-exact initial draws (model.py:458-605)."""
import warnings

import numpy as np
import pytest

from oracle import vimure_oracle as vo
from tests.golden_util import case_config, load_case
from vimure_amd.model import VimureModel


def small():
    d = load_case("A_ones_mut")
    return d["X"], d["R"]


def test_undirected_overrides_mutuality_with_warning():
    with pytest.warns(UserWarning, match="Overriding mutuality"):
        m = VimureModel(undirected=True, mutuality=True)
    assert m.mutuality is False


def test_undirected_needs_symmetric_x():
    X, R = small()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VimureModel(undirected=True)
    with pytest.raises(ValueError, match="has to be symmetric"):
        m.fit(X, R=R, K=2)


@pytest.mark.parametrize("kw,msg", [
    (dict(theta_prior=[0.1, 0.1]), "theta_prior must be a 2D tuple!"),
    (dict(lambda_prior=(1.0,)), "lambda_prior must be a 2D tuple!"),
    (dict(eta_prior=0.5), "eta_prior must be a 2D tuple!"),
    (dict(rho_prior=np.zeros((2, 3, 3))), "rho_prior has to have shape"),
    (dict(alpha_lambda=np.ones((3, 3)), beta_lambda=np.ones((1, 2))), "alpha_lambda matrix is not valid"),
    (dict(alpha_theta=np.ones((1, 3)), beta_theta=np.ones((1, 8))), "alpha_theta matrix is not valid"),
])
def test_bad_priors_raise_reference_messages(kw, msg):
    X, R = small()
    with pytest.raises(ValueError, match=msg):
        VimureModel().fit(X, R=R, K=2, **kw)


def test_bad_mask_shape():
    X, R = small()
    with pytest.raises(ValueError, match="Dimensions of reporter mask"):
        VimureModel().fit(X, R=R[:, :, :, :-1], K=2)


def test_counts_beyond_a_byte_become_coordinate_lists():
    """The reference holds int64 counts (utils.py:241-242): a dense array whose counts pass 255 reaches the engine as its
    coordinate lists (vmr_create_coo takes any count below 2^31); only what no layout holds is refused on the host."""
    from vimure_amd.tensor import engine_data, is_sparse_like
    X, R = small()
    big = engine_data(X.astype(np.int64) * 300)
    assert is_sparse_like(big) and int(big.vals.max()) == int(X.max()) * 300 and tuple(big.shape) == X.shape
    assert engine_data(X).dtype == np.uint8
    with pytest.raises(ValueError, match="2\\^31"):
        VimureModel().fit(X.astype(np.int64) * (2 ** 31), R=R, K=2)
    with pytest.raises(ValueError, match="M <= 8192"):
        engine_data(np.full((1, 2, 2, 9000), 300, np.int64))


def test_missing_k_and_r_warn_then_engine_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    X, R = small()
    with pytest.warns(UserWarning) as rec:
        with pytest.raises(Exception):      # validation passes, the engine then needs a HIP device
            VimureModel().fit(X, seed=1)
    msgs = " | ".join(str(w.message) for w in rec)
    assert "Parameter K was None. Defaulting to" in msgs and "Reporters Mask was not informed" in msgs


@pytest.mark.parametrize("name", ["A_ones_mut", "C_ones_nomut", "F_rho_prior", "E_undirected"])
def test_initial_draws_are_bit_exact(name):
    """The host class draws pr_rho / gamma / phi / nu exactly as the reference does for a seed."""
    d = load_case(name)
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VimureModel(mutuality=bool(d["mutuality"]), undirected=und)
        m._check_fit_params(d["X"], (10.0, 10.0), (0.1, 0.1), (0.5, 1.0), rho_prior, seed, R=d["R"], K=K, **priors)
    cov = (d["R"].any(axis=3) & (d["X"] != 0).any(axis=3)).astype(np.uint8)
    pr = m._draw_pr_rho(cov, 0.0)
    m._draw_gammas(float(d["X"].sum()))
    assert np.array_equal(pr, d["init_pr_rho"])
    for n in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte"):
        assert np.array_equal(np.broadcast_to(getattr(m, n), d["init_" + n].shape), d["init_" + n]), n
    assert m.nu_shp == float(d["init_nu_shp"]) and m.nu_rte == float(d["init_nu_rte"])


@pytest.mark.parametrize("L,N,K,bias,seed", [(1, 9, 2, 0.0, 1), (2, 21, 3, 0.2, 7), (1, 30, 5, 0.0, 12345), (2, 600, 2, 0.0, 4)])
def test_one_pass_draw_is_bit_identical_to_the_numpy_statements(L, N, K, bias, seed):
    """vimure_amd/csrc/host_init.c against the reference's statements (model.py:470-482, 536-556): same doubles,
    same RandomState stream afterwards (the gamma draws and the next seed follow it)."""
    from vimure_amd import _hostlib
    if _hostlib.load() is None:
        pytest.skip("no C compiler for the host helper")
    a, b = np.random.RandomState(seed), np.random.RandomState(seed)
    a.random_sample(5), b.random_sample(5)   # a generator state in the middle of a block
    cov = (np.random.RandomState(3).rand(L, N, N) < 0.6).astype(np.uint8)
    pr = 1.0 + 0.01 * a.rand(L, N, N, K)
    pr[..., 0] += bias
    pr /= pr.sum(axis=-1)[..., None]
    onehot = np.zeros(K)
    onehot[0] = 1.0
    pr[cov == 0] = onehot
    out = np.empty((L, N, N, K))
    got = _hostlib.draw_pr_rho(b, (L, N, N, K), bias, cov, out=out)
    assert got is not None and np.array_equal(got, pr) and np.shares_memory(got, out)
    assert np.array_equal(a.random_sample((L, 7)), b.random_sample((L, 7)))
    assert a.randint(1, 500) == b.randint(1, 500)


def test_initial_states_follow_the_reference_seed_chain():
    """`_initial_states` (drawn ahead of the GPU loop) yields what realisation-by-realisation draws yield."""
    from vimure_amd.model import VimureModel

    class FakeEngine:
        def __init__(self, shape):
            self.bufs = {}
            self.shape = shape

        def staging(self, i):
            return self.bufs.setdefault(i, np.empty(self.shape))

    L, N, M, K = 2, 11, 6, 3
    cov = (np.random.RandomState(5).rand(L, N, N) < 0.8).astype(np.uint8)

    def mk():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = VimureModel()
        m.L, m.N, m.M, m.K = L, N, M, K
        m.alpha_theta, m.beta_theta, m.alpha_lambda, m.beta_lambda = 0.1, 0.1, 10.0, 10.0
        m.alpha_mutuality, m.beta_mutuality, m.rho_prior = 0.5, 1.0, None
        m.num_realisations, m.bias0, m.sumX = 7, 0.3, 123.0
        m._change_seed(11)
        return m
    a, b = mk(), mk()
    got = [(r, s, {k: np.copy(v) for k, v in st.items()}, nxt) for r, s, st, nxt in a._initial_states(FakeEngine((L, N, N, K)), cov)]
    for r in range(7):   # the reference's order: prior, gammas, randint, reseed (model.py:386-437)
        bias = 0.0 if r < 5 else (r - 4) * 0.3
        pr = b._draw_pr_rho(cov, bias)
        st = b._draw_gammas(123.0)
        assert got[r][0] == r and got[r][1] == b.seed
        assert np.array_equal(got[r][2]["pr_rho"], pr)
        for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp"):
            assert np.array_equal(got[r][2][k], st[k])
        b._change_seed(b.seed + b.prng.randint(1, 500))
        assert got[r][3] == b.seed


def test_karnataka_tables_match_the_reference_loops():
    """`karnataka_tables` (vectorised) against a literal restatement of the reference driver's per-dyad / per-node loops
    (notebooks/python/experiments/karnataka.py:200-318) on a fitted-model stand-in."""
    import pandas as pd
    from vimure_amd.batch import karnataka_tables
    g = np.random.RandomState(0)
    N = 9
    X = np.zeros((1, N, N, N), np.uint8)
    R = np.zeros((1, N, N, N), np.uint8)
    rep = [0, 1, 2, 4, 5, 7]
    for i in range(N):
        for j in range(N):
            for m in (i, j):
                if m in rep and i != j:
                    R[0, i, j, m] = 1
                    X[0, i, j, m] = g.rand() < 0.3

    class M:
        pass
    m = M()
    m.num_realisations, m.max_iter, m.seed, m.maxL, m.G_exp_nu = 5, 101, 321, -12.5, 0.4
    m.G_exp_lambda_f, m.G_exp_theta_f = np.array([[0.01, 1.3]]), g.rand(1, N)
    m.rho_f = g.rand(1, N, N, 2)
    m.rho_f /= m.rho_f.sum(-1)[..., None]
    m.trace = pd.DataFrame({"realisation": [0], "seed": [3], "iter": [10], "elbo": [-1.0], "runtime": [0.1], "reached_convergence": [False]})
    thr = 0.54 * m.G_exp_nu - 0.01
    m.get_inferred_model = lambda method: (m.rho_f[..., 1] >= thr).astype(int)
    t = karnataka_tables(m, X, R, "vilA", "money", 3, 1.5)
    sumX = X.astype(int).sum(axis=3)
    union, inter, Yv = sumX > 0, sumX == 2, m.rho_f[..., 1] >= thr
    rows = []
    for i in range(N):
        for j in range(N):
            r = {"village": "vilA", "layer": "money", "initial_seed": 3, "source": i, "target": j, "dyad_ID": f"{i}_{j}",
                 "source_report": X[0, i, j, i] == 1, "target_report": X[0, i, j, j] == 1,
                 "vimure_posterior_probability": m.rho_f[0, i, j, 1], "in_union": union[0, i, j], "in_intersection": inter[0, i, j],
                 "in_vimure": Yv[0, i, j], "reciprocated_in_union": union[0, j, i], "reciprocated_in_intersection": inter[0, j, i],
                 "reciprocated_in_vimure": Yv[0, j, i]}
            if r["in_union"] or r["in_intersection"] or r["in_vimure"]:
                rows.append(r)
    ref = pd.DataFrame(rows)
    assert list(t["edgelist"].columns) == list(ref.columns) and len(ref) > 0
    pd.testing.assert_frame_equal(t["edgelist"].reset_index(drop=True), ref, check_dtype=False)
    rel = t["reliability"]
    assert list(rel.columns) == ["village", "layer", "initial_seed", "node", "theta", "lambda_theta", "is_node_reporter"]
    assert rel["is_node_reporter"].tolist() == [n in rep for n in range(N)]
    np.testing.assert_allclose(rel["lambda_theta"], 1.3 * m.G_exp_theta_f[0])
    s = t["summary"]
    assert list(s.columns) == ["running_time", "num_realisations", "max_iter", "initial_seed", "best_seed", "best_elbo", "eta_est",
                               "lambda_k", "model", "village", "layer"] and len(s) == 1 and s["lambda_k"][0] == [0.01, 1.3]
    assert list(t["trace"].columns[-3:]) == ["model", "village", "layer"]


def test_layer_draw_skips_the_other_layers_of_the_stream():
    """A rank of a layer-sharded fit draws only its own layers of pr_rho (reference model.py:470-482: one
    `rand(L, N, N, K)`): the generator is skipped over the others, and ends where the full draw ends."""
    from vimure_amd import _hostlib
    if _hostlib.load() is None:
        pytest.skip("no C compiler for the host helper")
    L, N, K = 6, 41, 3
    cov = (np.random.RandomState(1).rand(L, N, N) < 0.9).astype(np.uint8)
    ref = np.random.RandomState(11)
    full = 1.0 + 0.01 * ref.rand(L, N, N, K)
    full[..., 0] += 0.5
    full /= full.sum(axis=-1)[..., None]
    onehot = np.zeros(K)
    onehot[0] = 1.0
    full[cov == 0] = onehot
    after = ref.random_sample(4)
    for layers in ([0], [2, 5], [1, 2, 3], list(range(L))):
        g = np.random.RandomState(11)
        part = _hostlib.draw_pr_rho_layers(g, L, N, K, 0.5, layers, cov[layers])
        assert np.array_equal(part, full[layers]), layers
        assert np.array_equal(g.random_sample(4), after), layers
    g = np.random.RandomState(3)
    h = np.random.RandomState(3)
    assert _hostlib.mt_skip(g, 1000)
    h.random_sample(1000)
    assert np.array_equal(g.random_sample(5), h.random_sample(5))
    # K = 8: NumPy sums the 8 terms pairwise, the C pass does not -- the helper steps aside (NumPy statements run)
    assert _hostlib.draw_pr_rho(np.random.RandomState(0), (1, 5, 5, 8), 0.0, None) is None
