"""GPU: the lockstep loop of many small fits (vmr_fit_loop_batch: one launch per kernel and sweep for all handles) gives every
handle exactly what its own vmr_fit_loop gives (reference model.py:405-426, 1021-1056 per fit)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _village(N, seed, K=2, L=1):
    from vimure_amd.synthetic import standard_sbm
    from vimure_amd.tensor import SparseTensor
    net = standard_sbm(N=N, M=N, L=L, K=K, avg_degree=3.0, eta=0.3, seed=seed, flag_self_reporter=True)
    return SparseTensor.fromarray(net.X), SparseTensor.fromarray(net.R)


def _engine_with_state(X, R, K, seed, mutuality=True):
    from bench import draw_state
    from vimure_amd import CaviEngine
    eng = CaviEngine.from_coo(X.subs, X.vals, X.shape, R=R.subs, K=K, mutuality=mutuality)
    sum_x, cov = eng.data_stats()
    host, pr = draw_state(dict(L=int(X.shape[0]), N=int(X.shape[1]), M=int(X.shape[3]), K=K, mutuality=mutuality), seed, sum_x, cov)
    eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
    eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
    return eng


@pytest.mark.parametrize("K,mutuality", [(2, True), (3, True), (2, False)])
def test_lockstep_loop_equals_the_single_loops(K, mutuality):
    from vimure_amd import CaviEngine
    sizes = [40, 64, 90, 64, 33]
    data = [_village(N, s, K=K) for s, N in enumerate(sizes)]
    single, states = [], []
    for s, (X, R) in enumerate(data):
        eng = _engine_with_state(X, R, K, 10 + s, mutuality)
        single.append(eng.fit_loop(41, 0.1, 1))
        states.append(eng.get_state(rho=True))
        eng.close()
    engs = [_engine_with_state(X, R, K, 10 + s, mutuality) for s, (X, R) in enumerate(data)]
    both = CaviEngine.fit_loop_batch(engs, 41, 0.1, 1)
    for s, (eng, one, two) in enumerate(zip(engs, single, both)):
        assert [r[0] for r in one[0]] == [r[0] for r in two[0]] and one[2] == two[2] and one[3] == two[3], s
        np.testing.assert_allclose([r[1] for r in two[0]], [r[1] for r in one[0]], rtol=1e-10)
        assert abs(one[1] - two[1]) <= 1e-10 * abs(one[1])
        st = eng.get_state(rho=True)
        for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "rho"):
            np.testing.assert_allclose(st[k], states[s][k], rtol=1e-9, atol=1e-12, err_msg=f"{s} {k}")
        eng.close()


def test_lockstep_with_a_handle_of_another_kind_and_early_convergence():
    """A K = 3 handle among K = 2 ones runs its own loop; a tolerance that stops some fits early takes them out of the launch."""
    from vimure_amd import CaviEngine
    data = [(_village(50, 1), 2), (_village(48, 2, K=3), 3), (_village(70, 3), 2), (_village(36, 4), 2)]
    single = []
    for s, ((X, R), K) in enumerate(data):
        eng = _engine_with_state(X, R, K, 20 + s)
        single.append(eng.fit_loop(101, 5.0, 0))
        eng.close()
    engs = [_engine_with_state(X, R, K, 20 + s) for s, ((X, R), K) in enumerate(data)]
    both = CaviEngine.fit_loop_batch(engs, 101, 5.0, 0)
    assert len({r[2] for r in single}) > 1   # (the fits stop at different iterations)
    for one, two, eng in zip(single, both, engs):
        assert one[2] == two[2] and one[3] == two[3]
        assert abs(one[1] - two[1]) <= 1e-10 * abs(one[1])
        eng.close()


def test_fit_datasets_lockstep_equals_threads():
    from vimure_amd.batch import fit_datasets
    data = {f"v{i}": _village(N, i, L=2) for i, N in enumerate([40, 56])}
    a = fit_datasets(data, K=2, seeds=[1, 2], num_realisations=2, max_iter=31, lockstep=True)
    b = fit_datasets(data, K=2, seeds=[1, 2], num_realisations=2, max_iter=31, lockstep=False, workers=1)
    assert a[["dataset", "layer", "seed", "iters"]].values.tolist() == b[["dataset", "layer", "seed", "iters"]].values.tolist()
    np.testing.assert_allclose(a["elbo"].values, b["elbo"].values, rtol=1e-10)
    np.testing.assert_allclose(a["nu"].values, b["nu"].values, rtol=1e-9)


def test_lockstep_with_several_layers_and_five_categories():
    """Units of L = 2 layers and K = 5 (the 512-thread lockstep variants), all-ones mask next to self-reporter masks of the same kind
    only: the all-ones unit runs alone, the others together."""
    from vimure_amd import CaviEngine
    from vimure_amd.synthetic import standard_sbm
    from vimure_amd.tensor import SparseTensor
    K = 5
    data = [_village(48, 1, K=K, L=2), _village(60, 2, K=K, L=2), _village(36, 3, K=K, L=2)]
    net = standard_sbm(N=40, M=12, L=2, K=K, avg_degree=4.0, eta=0.4, seed=9)
    single = []
    for s, (X, R) in enumerate(data):
        eng = _engine_with_state(X, R, K, 30 + s)
        single.append((eng.fit_loop(31, 0.1, 1), eng.get_state(rho=True)))
        eng.close()
    engs = [_engine_with_state(X, R, K, 30 + s) for s, (X, R) in enumerate(data)]
    both = CaviEngine.fit_loop_batch(engs, 31, 0.1, 1)
    for (one, st1), two, eng in zip(single, both, engs):
        assert [r[0] for r in one[0]] == [r[0] for r in two[0]] and one[2] == two[2]
        np.testing.assert_allclose([r[1] for r in two[0]], [r[1] for r in one[0]], rtol=1e-10)
        st2 = eng.get_state(rho=True)
        for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "rho"):
            np.testing.assert_allclose(st2[k], st1[k], rtol=1e-9, atol=1e-12, err_msg=k)
        eng.close()
    del net


def _oracle_loop(c, max_iter, tol, decision):
    """The reference's stop rule (model.py:1021-1056) around the coordinate-list oracle: ELBO at iteration 1, every 10th, the last."""
    it, coincide, reached, elbo, rows = 1, 0, False, -1e10, []
    while not reached and it <= max_iter:
        c.cavi_step()
        if it == 1 or it % 10 == 0 or it == max_iter:
            old, elbo = elbo, c.elbo()
            coincide = coincide + 1 if abs(elbo - old) < tol else 0
            reached = coincide > decision
        it += 1
        if (it - 1) % 10 == 0:
            rows.append((it - 1, elbo))
    return rows, elbo, it - 1, reached


def _against_oracle(units, K, max_iter):
    """units: [(X, R)] coordinate containers -> lockstep loop of all of them, each held to cavi_coo from the same drawn state."""
    from bench import draw_state
    from oracle import cavi_coo
    from vimure_amd import CaviEngine
    engs, refs = [], []
    for s, (X, R) in enumerate(units):
        L, N, M = int(X.shape[0]), int(X.shape[1]), int(X.shape[3])
        eng = CaviEngine.from_coo(X.subs, X.vals, X.shape, R=R.subs, K=K, mutuality=True)
        sum_x, cov = eng.data_stats()
        host, pr = draw_state(dict(L=L, N=N, M=M, K=K, mutuality=True), 40 + s, sum_x, cov)
        eng.set_priors(0.1, 0.1, 10.0, 10.0, 0.5, 1.0)
        eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
        engs.append(eng)
        refs.append(cavi_coo.CooRef((X.subs, np.asarray(X.vals).astype(np.int32)), R.subs, X.shape, K, True, (0.1, 0.1, 10.0, 10.0, 0.5, 1.0),
                                    host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr))
    res = CaviEngine.fit_loop_batch(engs, max_iter, 0.1, 1)
    for eng, c, (rows, elbo, its, conv) in zip(engs, refs, res):
        o_rows, o_elbo, o_its, o_conv = _oracle_loop(c, max_iter, 0.1, 1)
        assert its == o_its and conv == o_conv and [r[0] for r in rows] == [r[0] for r in o_rows]
        np.testing.assert_allclose([r[1] for r in rows], [r[1] for r in o_rows], rtol=1e-9)
        assert abs(elbo - o_elbo) <= 1e-9 * abs(o_elbo)
        st = eng.get_state(rho=True)
        np.testing.assert_allclose(st["rho"], c.rho, rtol=1e-7, atol=1e-12)
        for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte"):
            np.testing.assert_allclose(st[k], getattr(c, k), rtol=1e-9, err_msg=k)
        assert abs(st["nu_shp"] - c.nu_shp) <= 1e-9 * abs(c.nu_shp)
        eng.close()


def test_lockstep_at_the_largest_village_against_the_oracle():
    """BASELINE configs[3]'s upper sizes through the lockstep launch: the largest Karnataka village (N = 898 nodes, 413 respondents,
    reporter dimension N, self-reporter mask, L = 1) beside a median and the smallest one, every unit against oracle/cavi_coo.c."""
    from bench import village_coo
    from vimure_amd.tensor import layer_of
    units = []
    for s, (N, nresp) in enumerate([(898, 413), (447, 217), (198, 94)]):
        X, R = village_coo(N, 50 + s, "cuda:0", L=1, n_resp=nresp)
        units.append((layer_of(X, 0), layer_of(R, 0)))
    _against_oracle(units, 2, 21)


def test_lockstep_three_categories_two_layers_against_the_oracle():
    """K = 3, L = 2 units in one lockstep launch, each against the oracle (not only against its own single loop)."""
    _against_oracle([_village(N, 60 + s, K=3, L=2) for s, N in enumerate([44, 57, 70])], 3, 31)


def test_groups_on_several_lanes_give_the_same_fits():
    """Groups of `width` units worked on by several host threads at once (fit_units_lockstep(lanes=...): one group's host share
    beside another's lockstep loop) give, fit by fit, what one lane gives."""
    from vimure_amd.batch import fit_units_lockstep
    units = [(i,) + _village(N, 70 + i) for i, N in enumerate([40, 52, 64, 45, 58])]
    kw = dict(num_realisations=2, max_iter=21)
    one = fit_units_lockstep(units, 2, [1, 2], True, None, kw, width=2, lanes=1)
    three = fit_units_lockstep(units, 2, [1, 2], True, None, kw, width=2, lanes=3)
    assert sorted(one) == sorted(three) == list(range(5))
    for tag in one:
        a, b = one[tag][0], three[tag][0]
        assert [(r["seed"], r["iters"], r["converged"]) for r in a] == [(r["seed"], r["iters"], r["converged"]) for r in b]
        np.testing.assert_allclose([r["elbo"] for r in b], [r["elbo"] for r in a], rtol=1e-10)
        np.testing.assert_allclose([r["nu"] for r in b], [r["nu"] for r in a], rtol=1e-9)


@pytest.mark.parametrize("K,L", [(2, 1), (3, 2)])
def test_graph_replay_of_sweeps_equals_the_eager_loop(K, L, monkeypatch):
    """VMR_GRAPH=1: the plain sweeps of vmr_step / the fit loop replayed as hipGraphs (up to 9 per launch, rho unwritten inside
    them) give the trace, the ELBO and the state of the same loop queued launch by launch -- and, across two realisations on one
    handle, graphs do not outlive vmr_set_state."""
    X, R = _village(70, 91, K=K, L=L)
    res = {}
    for mode in ("eager", "graph"):
        if mode == "graph":
            monkeypatch.setenv("VMR_GRAPH", "1")
        eng = _engine_with_state(X, R, K, 17)
        a = eng.fit_loop(41, 0.1, 1)
        eng.step(13)
        e1 = eng.step(7, want_elbo=True)
        st1 = eng.get_state(rho=True)
        # a second realisation on the same handle: another state, the same calls
        from bench import draw_state
        sum_x, cov = eng.data_stats()
        host, pr = draw_state(dict(L=int(X.shape[0]), N=int(X.shape[1]), M=int(X.shape[3]), K=K, mutuality=True), 18, sum_x, cov)
        eng.set_state(host.gamma_shp, host.gamma_rte, host.phi_shp, host.phi_rte, host.nu_shp, host.nu_rte, pr)
        b = eng.fit_loop(31, 0.1, 1)
        st2 = eng.get_state(rho=True)
        res[mode] = (a, e1, st1, b, st2)
        eng.close()
    (a0, e0, s0, b0, t0), (a1, e1, s1, b1, t1) = res["eager"], res["graph"]
    for x, y in ((a0, a1), (b0, b1)):
        assert [r[0] for r in x[0]] == [r[0] for r in y[0]] and x[2] == y[2] and x[3] == y[3]
        np.testing.assert_allclose([r[1] for r in y[0]], [r[1] for r in x[0]], rtol=1e-10)
    assert abs(e0 - e1) <= 1e-10 * abs(e0)
    for u, v in ((s0, s1), (t0, t1)):
        for k in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "rho"):
            np.testing.assert_allclose(v[k], u[k], rtol=1e-9, atol=1e-12, err_msg=k)
