"""GPU parity at sizes the small golden cases do not reach (M = 200 rows, 13 chunks per row, heavy
true-tie rows, several tile pairs per workgroup, ragged edge tiles): HIP engine vs the plain-C oracle
on the same seeded synthetic network after a few sweeps, plus size-independent properties."""
import numpy as np
import pytest

from oracle import cavi_ref
from oracle import vimure_oracle as vo

pytestmark = pytest.mark.gpu


def _run(L, N, M, K, eta, mutuality, mask, sweeps=3, seed=2, fmt=None):
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    net = standard_sbm(N=N, M=M, L=L, K=K, avg_degree=5.0, eta=eta, seed=0)
    g = np.random.RandomState(7)
    if mask == "ones":
        R = None
    elif mask == "random":
        R = (g.rand(L, N, N, M) < 0.7).astype(np.uint8)
    elif mask == "mixed":   # rows that are all ones (summed by the rho pass), empty, and partial (mask kernel)
        kind = g.randint(0, 3, size=(L, N, N, 1))
        R = np.where(kind == 0, 1, np.where(kind == 1, 0, g.rand(L, N, N, M) < 0.5)).astype(np.uint8)
    X = net.X
    pr = vo.make_priors(L, M, K)
    pb_cov = (np.ones((L, N, N), bool) if R is None else R.any(axis=3)) & (X != 0).any(axis=3)
    prng = np.random.RandomState(seed)
    pr_rho = 1.0 + 0.01 * prng.rand(L, N, N, K)
    pr_rho /= pr_rho.sum(axis=-1)[..., None]
    onehot = np.zeros(K); onehot[0] = 1.0
    pr_rho[~pb_cov] = onehot
    gs = 0.1 * prng.random_sample((L, M)) + 0.1
    ps = 10.0 * prng.random_sample((L, K)) + 10.0
    gr = 0.1 * prng.random_sample((L, M)) + 0.1
    prt = 10.0 * prng.random_sample((L, K)) + 10.0
    nu_s, nu_r = (0.5 * prng.random_sample(1)[0] + 0.5, 1.0 + float(X.sum())) if mutuality else (1e-6, 1.0)
    c = cavi_ref.CRef(X, R, K, mutuality, (0.1, 0.1, 10.0, 10.0, 0.5, 1.0), gs, gr, ps, prt, nu_s, nu_r, pr_rho)
    eng = CaviEngine(X, R, K=K, mutuality=mutuality)
    if fmt is not None:
        assert eng.data_format()[0] == fmt   # the layout under test is the one that runs
    s, cov = eng.data_stats()
    assert s == float(X.sum()) and np.array_equal(cov.astype(bool), pb_cov)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    eng.set_state(gs, gr, ps, prt, nu_s, nu_r, pr_rho)
    e_gpu = None
    for it in range(sweeps):
        c.cavi_step()
        e_gpu = eng.step(1, want_elbo=True)
    e_cpu = c.elbo()
    st = eng.get_state()
    assert abs(e_gpu - e_cpu) <= 1e-9 * max(1.0, abs(e_cpu)), (e_gpu, e_cpu)
    assert abs(eng.elbo() - e_cpu) <= 1e-9 * max(1.0, abs(e_cpu))
    np.testing.assert_allclose(st["rho"], c.rho, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(st["gamma_shp"], c.gamma_shp, rtol=1e-9)
    np.testing.assert_allclose(st["gamma_rte"], c.gamma_rte, rtol=1e-9)
    np.testing.assert_allclose(st["phi_shp"], c.phi_shp, rtol=1e-9)
    np.testing.assert_allclose(st["phi_rte"], c.phi_rte, rtol=1e-9)
    np.testing.assert_allclose(st["nu_shp"], c.nu_shp, rtol=1e-9)
    # properties that hold at any size: rho rows are distributions (or all-zero after underflow), finite
    rs = st["rho"].sum(axis=-1)
    assert np.all(np.isfinite(st["rho"])) and np.all((np.abs(rs - 1.0) < 1e-12) | (rs == 0.0))
    eng.close()


def test_m200_mutuality_all_ones_mask(vmr_format):
    _run(L=2, N=333, M=200, K=2, eta=0.5, mutuality=True, mask="ones", fmt=vmr_format)     # ragged tiles: 333 = 41*8 + 5


def test_m200_random_mask_k3(vmr_format):
    _run(L=1, N=250, M=200, K=3, eta=0.4, mutuality=True, mask="random", fmt=vmr_format)


def test_m1000_tile_edge4_no_mutuality(vmr_format):
    _run(L=1, N=90, M=1000, K=2, eta=0.0, mutuality=False, mask="random", sweeps=2, fmt=vmr_format)   # b = 4, 16 mask words


def test_mixed_mask_rows_full_empty_partial(vmr_format):
    _run(L=2, N=120, M=100, K=2, eta=0.5, mutuality=True, mask="mixed", sweeps=4, fmt=vmr_format)


def test_m2000_tile_edge2(vmr_format):
    _run(L=1, N=40, M=2000, K=2, eta=0.5, mutuality=True, mask="random", sweeps=2, fmt=vmr_format)     # b = 2, 32 lanes per tie


def test_m7000_tile_edge1_chunk_walk(vmr_format):
    _run(L=1, N=20, M=7000, K=2, eta=0.0, mutuality=False, mask="ones", sweeps=2, fmt=vmr_format)      # b = 1, 64 lanes per tie


def test_m4000_mutuality_tables_fill_lds(vmr_format):
    _run(L=1, N=24, M=4000, K=2, eta=0.4, mutuality=True, mask="random", sweeps=2, fmt=vmr_format)     # ~150 KB of LDS tables


def test_reporter_tables_beyond_lds_are_refused(monkeypatch):
    """Dense tiles keep (K + 2) * 8 bytes per reporter in LDS and refuse what does not fit; report lists keep as many
    table levels as fit and read the rest from global memory, so the same shape is accepted there."""
    from vimure_amd.engine import CaviEngine
    X = np.zeros((1, 4, 4, 7000), np.uint8)
    monkeypatch.setenv("VMR_FORMAT", "dense")
    with pytest.raises(ValueError, match="LDS"):
        CaviEngine(X, None, K=2, mutuality=True)
    monkeypatch.setenv("VMR_FORMAT", "sparse")
    _run(L=1, N=12, M=7000, K=2, eta=0.4, mutuality=True, mask="random", sweeps=2, fmt="sparse")


def test_m8300_beyond_the_list_format_uses_dense_tiles(monkeypatch):
    """Report lists hold the reporter in 13 bits (M <= 8192); wider tensors stay on the dense tiles."""
    monkeypatch.delenv("VMR_FORMAT", raising=False)
    _run(L=1, N=10, M=8300, K=2, eta=0.0, mutuality=False, mask="random", sweeps=2, fmt="dense")


def test_wide_self_reporter_mask_falls_back_to_mask_words(monkeypatch):
    """Mask lists need A[Mp][K] in LDS (k_mask_lists); M = 3000 with K = 8 does not fit, the mask words take over."""
    from vimure_amd import CaviEngine
    monkeypatch.setenv("VMR_FORMAT", "sparse")
    N = M = 3000
    X = np.zeros((1, 6, 6, M), np.uint8)   # N must equal M only for the mask's meaning, not for the engine
    R = np.zeros((1, 6, 6, M), np.uint8)
    for i in range(6):
        for j in range(6):
            R[0, i, j, [i, j]] = 1
            if i != j:
                X[0, i, j, i] = 1 + (i + j) % 3
    eng = CaviEngine(X, R, K=8, mutuality=False)
    assert eng.mask_format()[0] == "words"
    g = np.random.RandomState(0)
    pr = 1.0 + 0.01 * g.rand(1, 6, 6, 8)
    pr /= pr.sum(-1)[..., None]
    init = (0.1 + 0.1 * g.rand(1, M), 0.1 + 0.1 * g.rand(1, M), 10 + 10 * g.rand(1, 8), 10 + 10 * g.rand(1, 8), 1e-6, 1.0, pr)
    c = cavi_ref.CRef(X, R, 8, False, (0.1, 0.1, 10.0, 10.0, 0.5, 1.0), *init)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    eng.set_state(*init)
    e = eng.step(2, want_elbo=True)
    c.cavi_step(); c.cavi_step()
    assert abs(e - c.elbo()) <= 1e-9 * abs(c.elbo())
    np.testing.assert_allclose(eng.get_state()["rho"], c.rho, rtol=1e-8, atol=1e-13)
    eng.close()


def test_k5_categories(vmr_format):
    _run(L=1, N=150, M=64, K=5, eta=0.5, mutuality=True, mask="mixed", sweeps=3, fmt=vmr_format)


def test_m1000_k3_mutuality_two_lds_levels(vmr_format):
    _run(L=1, N=60, M=1000, K=3, eta=0.5, mutuality=True, mask="ones", sweeps=2, fmt=vmr_format)       # H levels 0-1 in LDS beside the rho pass


def test_m1500_k4_statistics_in_a_second_pass(vmr_format):
    _run(L=1, N=40, M=1500, K=4, eta=0.5, mutuality=True, mask="random", sweeps=2, fmt=vmr_format)     # tables too wide: two passes per sweep


def test_dense_and_report_list_formats_agree(monkeypatch):
    """The two data layouts of the engine (vmr_data_format) are two implementations of the same sweep."""
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    L, N, M, K = 2, 96, 120, 2
    net = standard_sbm(N=N, M=M, L=L, K=K, avg_degree=5.0, eta=0.5, seed=3)
    g = np.random.RandomState(5)
    kind = g.randint(0, 3, size=(L, N, N, 1))
    R = np.where(kind == 0, 1, np.where(kind == 1, 0, g.rand(L, N, N, M) < 0.5)).astype(np.uint8)
    pr = 1.0 + 0.01 * g.rand(L, N, N, K)
    pr /= pr.sum(-1)[..., None]
    init = (0.1 + 0.1 * g.rand(L, M), 0.1 + 0.1 * g.rand(L, M), 10 + 10 * g.rand(L, K), 10 + 10 * g.rand(L, K), 0.7,
            1.0 + float(net.X.sum()), pr)
    out = {}
    for fmt in ("dense", "sparse"):
        monkeypatch.setenv("VMR_FORMAT", fmt)
        eng = CaviEngine(net.X, R, K=K, mutuality=True)
        assert eng.data_format() == (fmt, int((net.X != 0).sum()) if fmt == "sparse" else eng.data_format()[1])
        eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
        eng.set_state(*init)
        e = [eng.step(1, want_elbo=True) for _ in range(4)]
        out[fmt] = (e, eng.elbo(), eng.get_state())
        eng.close()
    (ed, ed2, sd), (es, es2, ss) = out["dense"], out["sparse"]
    np.testing.assert_allclose(es, ed, rtol=1e-11)
    assert abs(es2 - ed2) <= 1e-11 * abs(ed2)
    for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp"):   # (four sweeps: differences of the last bits of rho have grown to 1e-11)
        np.testing.assert_allclose(ss[k], sd[k], rtol=1e-10)
    np.testing.assert_allclose(ss["rho"], sd["rho"], rtol=1e-9, atol=1e-13)


def test_self_reporter_mask_lists_agree_with_mask_words(monkeypatch):
    """Karnataka-shaped mask (R = 1 iff the reporter is one end of the tie): partial rows are walked as reporter
    lists; the same fit with the lists disabled reads the bit-packed words."""
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    L, N, K = 2, 280, 2   # 5 mask words per row against a 2-reporter list
    net = standard_sbm(N=N, M=N, L=L, K=K, avg_degree=3.0, eta=0.3, seed=4, flag_self_reporter=True)
    g = np.random.RandomState(6)
    pr = 1.0 + 0.01 * g.rand(L, N, N, K)
    pr /= pr.sum(-1)[..., None]
    init = (0.1 + 0.1 * g.rand(L, N), 0.1 + 0.1 * g.rand(L, N), 10 + 10 * g.rand(L, K), 10 + 10 * g.rand(L, K), 0.7,
            1.0 + float(net.X.sum()), pr)
    c = cavi_ref.CRef(net.X, net.R, K, True, (0.1, 0.1, 10.0, 10.0, 0.5, 1.0), init[0], init[1], init[2], init[3],
                      init[4], init[5], pr)
    for _ in range(4):
        c.cavi_step()
    e_cpu = c.elbo()
    out = {}
    monkeypatch.setenv("VMR_FORMAT", "sparse")   # mask lists belong to the report-list format
    for mode in ("lists", "words"):
        if mode == "words":
            monkeypatch.setenv("VMR_NO_RLISTS", "1")
        eng = CaviEngine(net.X, net.R, K=K, mutuality=True)
        assert eng.mask_format() == (mode, int(net.R.sum()) if mode == "lists" else 0)
        eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
        eng.set_state(*init)
        e = eng.step(4, want_elbo=True)
        st = eng.get_state()
        assert abs(e - e_cpu) <= 1e-9 * abs(e_cpu)
        np.testing.assert_allclose(st["gamma_rte"], c.gamma_rte, rtol=1e-9)
        np.testing.assert_allclose(st["phi_rte"], c.phi_rte, rtol=1e-9)
        np.testing.assert_allclose(st["rho"], c.rho, rtol=1e-7, atol=1e-12)
        out[mode] = (e, st)
        eng.close()
    assert abs(out["lists"][0] - out["words"][0]) <= 1e-11 * abs(e_cpu)


def test_m50_small_rows(vmr_format):
    _run(L=3, N=200, M=50, K=2, eta=0.5, mutuality=True, mask="ones", fmt=vmr_format)


def test_subnormal_normaliser_is_divided_not_inverted():
    """A tie whose unnormalised rho sums to a subnormal number must still normalise (rho/sum, model.py:811);
    1/sum would overflow.  Force it with a tie reported by everybody under a tiny lambda prior mass."""
    from vimure_amd import CaviEngine
    L, N, M, K = 1, 16, 200, 2
    X = np.zeros((L, N, N, M), np.uint8)
    X[0, 1, 2, :] = 3
    X[0, 2, 1, :] = 1
    pr_rho = np.full((L, N, N, K), 0.5)
    gs, gr = np.full((L, M), 0.2), np.full((L, M), 5.0)
    ps, prt = np.full((L, K), 1.0), np.full((L, K), 40.0)
    pri = (0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    c = cavi_ref.CRef(X, None, K, True, pri, gs, gr, ps, prt, 0.7, 1.0 + float(X.sum()), pr_rho)
    eng = CaviEngine(X, None, K=K, mutuality=True)
    eng.set_priors(*pri)
    eng.set_state(gs, gr, ps, prt, 0.7, 1.0 + float(X.sum()), pr_rho)
    from vimure_amd import _lib
    eng.sub_step(_lib.STEP_RHO)
    c.update_rho()
    g = eng.get_state()["rho"]
    assert np.all(np.isfinite(g))
    np.testing.assert_allclose(g, c.rho, rtol=1e-9, atol=1e-300)
    eng.close()


@pytest.mark.parametrize("mask", ["ones", "self"])
def test_rho_of_inner_sweeps_is_not_written_and_nothing_can_tell(mask, monkeypatch):
    """vmr_step(n) writes rho on its last sweep only (the rho of an inner sweep is overwritten unread: a dead store of a third of the
    pass' bytes).  n sweeps in one call == n calls of one sweep; and with VMR_DEBUG_LAZY_RHO=1 -- not even the last sweep writes
    -- every reader of rho (vmr_get_state, vmr_readout, vmr_elbo, vmr_snapshot) gets the same values through ensure_rho's re-write."""
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    L, N, K = 2, 150, 2
    M = N if mask == "self" else 20
    net = standard_sbm(N=N, M=M, L=L, K=K, avg_degree=4.0, eta=0.4, seed=13, flag_self_reporter=(mask == "self"))
    R = net.R if mask == "self" else None
    g = np.random.RandomState(2)
    pr = 1.0 + 0.01 * g.rand(L, N, N, K)
    pr /= pr.sum(-1)[..., None]
    init = (0.1 + 0.1 * g.rand(L, M), 0.1 + 0.1 * g.rand(L, M), 10 + 10 * g.rand(L, K), 10 + 10 * g.rand(L, K), 0.7, 1.0 + float(net.X.sum()), pr)

    def run(mode):
        eng = CaviEngine(net.X, R, K=K, mutuality=True)
        assert eng.data_format()[0] == "sparse"
        eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
        eng.set_state(*init)
        if mode == "one by one":
            for _ in range(6):
                eng.step(1)
        else:
            eng.step(6)
        out = (eng.readout("rho_mean"), eng.get_state(), eng.elbo())
        eng.snapshot(); eng.restore()
        out += (eng.get_state()["rho"],)
        eng.close()
        return out
    a = run("one by one")
    b = run("one call")
    monkeypatch.setenv("VMR_DEBUG_LAZY_RHO", "1")
    c = run("one call")
    for other in (b, c):
        np.testing.assert_allclose(other[0], a[0], rtol=1e-9, atol=1e-13)
        for k in ("rho", "gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp"):
            np.testing.assert_allclose(other[1][k], a[1][k], rtol=1e-9, atol=1e-13, err_msg=k)
        assert abs(other[2] - a[2]) <= 1e-10 * abs(a[2])
        np.testing.assert_allclose(other[3], a[1]["rho"], rtol=1e-9, atol=1e-13)
