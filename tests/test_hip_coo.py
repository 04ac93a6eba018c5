"""GPU: the coordinate-list entry point (vmr_create_coo): the reference's containers (`X.subs`, `X.vals`, `R.subs`)
go to the device as they are and become the report lists there -- no dense [L,N,N,M] tensor.  Held to the golden
vectors of the reference, to the dense entry point, and to the coordinate-list oracle (oracle/cavi_coo.c)."""
import warnings

import numpy as np
import pytest

from oracle import cavi_coo
from oracle import vimure_oracle as vo
from tests.golden_util import case_config, case_names, load_case

pytestmark = pytest.mark.gpu
PRI = (0.1, 0.1, 10.0, 10.0, 0.5, 1.0)


def _coo_engine(X, R, K, mut, implicit_ones=False):
    from vimure_amd import CaviEngine
    sx = np.nonzero(X)
    Rs = None if (R is None or implicit_ones) else np.nonzero(R)
    return CaviEngine.from_coo(sx, X[sx], X.shape, R=Rs, K=K, mutuality=mut)


@pytest.mark.parametrize("name", [n for n in case_names() if not n.startswith("H_ref_f1_over")])
def test_coo_substeps_match_oracle_and_golden(name):
    from vimure_amd import _lib
    d = load_case(name)
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    L, N, _, M = d["X"].shape
    pr = vo.make_priors(L, M, K, **priors)
    pb = vo.Problem(d["X"], d["R"], K, mut, pr, undirected=und)
    all_ones = bool(np.all(d["R"] == 1))
    for implicit in ([False, True] if all_ones else [False]):
        st = vo.init_state(pb, np.random.RandomState(seed), rho_prior=rho_prior)
        eng = _coo_engine(d["X"], d["R"], K, mut, implicit_ones=implicit)
        assert eng.data_format() == ("sparse", int((d["X"] != 0).sum()))
        s, cov = eng.data_stats()
        assert s == pb.sumX and np.array_equal(cov, (pb.R.any(axis=3) & (pb.X != 0).any(axis=3)).astype(np.uint8))
        eng.set_priors(pr.alpha_theta, pr.beta_theta, pr.alpha_lambda, pr.beta_lambda, pr.alpha_eta, pr.beta_eta)
        eng.set_state(st.gamma_shp, st.gamma_rte, st.phi_shp, st.phi_rte, st.nu_shp, st.nu_rte, st.pr_rho)
        for it in range(1, len(d["step_elbo"]) + 1):
            eng.sub_step(_lib.STEP_GAMMA)
            eng.sub_step(_lib.STEP_PHI)
            eng.sub_step(_lib.STEP_RHO)
            eng.sub_step(_lib.STEP_NU)
            g = eng.get_state()
            np.testing.assert_allclose(g["gamma_shp"], d[f"it{it}_gamma_shp"], rtol=1e-9)
            np.testing.assert_allclose(g["gamma_rte"], d[f"it{it}_gamma_rte"], rtol=1e-9)
            np.testing.assert_allclose(g["phi_rte"], d[f"it{it}_phi_rte"], rtol=1e-9)
            np.testing.assert_allclose(g["rho"], d[f"it{it}_rho"], rtol=1e-9, atol=1e-13)
            ref = float(d["step_elbo"][it - 1])
            assert abs(eng.elbo() - ref) <= 1e-9 * max(1.0, abs(ref))
        eng.close()


def test_coo_inputs_validated():
    from vimure_amd import CaviEngine
    sub = (np.array([0, 0]), np.array([1, 1]), np.array([2, 2]), np.array([3, 3]))
    with pytest.raises(ValueError, match="duplicate"):
        CaviEngine.from_coo(sub, np.array([1, 2]), (1, 5, 5, 4), K=2)
    with pytest.raises(ValueError, match="outside"):
        CaviEngine.from_coo((np.array([0]), np.array([5]), np.array([0]), np.array([0])), np.array([1]), (1, 5, 5, 4), K=2)
    with pytest.raises(ValueError, match="positive"):
        CaviEngine.from_coo((np.array([0]), np.array([1]), np.array([0]), np.array([0])), np.array([0]), (1, 5, 5, 4), K=2)
    eng = CaviEngine.from_coo((np.array([0]), np.array([1]), np.array([0]), np.array([0])), np.array([2048]), (1, 5, 5, 4), K=2)
    assert eng.data_stats()[0] == 2048.0   # beyond the 11-bit count field of a packed entry: two-word entries, the general kernels
    eng.close()
    eng = CaviEngine.from_coo(tuple(np.zeros(0, np.int64) for _ in range(4)), np.zeros(0, np.int64), (1, 5, 5, 4), K=2)   # no reports at all
    assert eng.data_format() == ("sparse", 0) and eng.data_stats()[0] == 0.0
    eng.close()


def test_karnataka_shaped_village_from_lists():
    """A village layer as the reader produces it (self-reporter mask, M-dim = N): 2 (N - 1) mask entries per reporter,
    a few hundred reports -- from torch tensors on the GPU, against the coordinate-list oracle."""
    import torch
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    N, K = 420, 2
    net = standard_sbm(N=N, M=N, L=2, K=K, avg_degree=3.0, eta=0.3, seed=4, flag_self_reporter=True)
    sx, sr = np.nonzero(net.X), np.nonzero(net.R)
    eng = CaviEngine.from_coo(tuple(torch.as_tensor(a).cuda() for a in sx), torch.as_tensor(net.X[sx].astype(np.int64)).cuda(),
                              net.X.shape, R=tuple(torch.as_tensor(a).cuda() for a in sr), K=K, mutuality=True)
    assert eng.mask_format() == ("lists", int(net.R.sum()))
    g = np.random.RandomState(6)
    pr = 1.0 + 0.01 * g.rand(2, N, N, K)
    pr /= pr.sum(-1)[..., None]
    init = (0.1 + 0.1 * g.rand(2, N), 0.1 + 0.1 * g.rand(2, N), 10 + 10 * g.rand(2, K), 10 + 10 * g.rand(2, K), 0.7,
            1.0 + float(net.X.sum()), pr)
    c = cavi_coo.CooRef((sx, net.X[sx]), sr, net.X.shape, K, True, PRI, *init)
    eng.set_priors(*PRI)
    eng.set_state(*init)
    for _ in range(4):
        c.cavi_step()
    e = eng.step(4, want_elbo=True)
    assert abs(e - c.elbo()) <= 1e-9 * abs(c.elbo())
    st = eng.get_state()
    np.testing.assert_allclose(st["rho"], c.rho, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(st["gamma_rte"], c.gamma_rte, rtol=1e-9)
    np.testing.assert_allclose(st["nu_shp"], c.nu_shp, rtol=1e-9)
    eng.close()


def test_fit_accepts_the_readers_containers_without_densifying(monkeypatch):
    """`fit(DataFrame)` / `fit(sptensor-like)`: the reader's coordinate containers reach the engine as lists (the dense
    conversion is never called) and give the reference's fit (golden case I, Karnataka village 1 'money')."""
    import os
    import pandas as pd
    import vimure_amd.model as vmm
    from tests.golden_util import GOLDEN
    from vimure_amd import VimureModel

    def boom(*a, **k):
        raise AssertionError("dense conversion called on the coordinate-list path")
    monkeypatch.setattr(vmm, "to_dense_u8", boom)
    J = dict(np.load(os.path.join(GOLDEN, "J_edgelist_io.npz")))
    df = pd.DataFrame({c: J["vil1_df_" + c] for c in ("reporter", "ego", "alter", "weight", "layer")})
    d = load_case("I_karnataka_vil1_money")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VimureModel().fit(df, seed=1, num_realisations=1, max_iter=21)
    assert (m.L, m.N, m.M, m.K) == (1, 324, 324, 2)
    assert m.trace["iter"].tolist() == d["fit_trace_iter"].tolist()
    assert abs(m.maxL - float(d["fit_maxL"])) <= 1e-8 * abs(float(d["fit_maxL"]))
    np.testing.assert_allclose(m.rho_f, d["fit_rho_f"], rtol=1e-6, atol=1e-12)
