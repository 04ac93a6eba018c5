"""GPU: posterior read-out (`get_inferred_model`, `predict`, `sample_inferred_model`, `get_posterior_estimates`;
reference model.py:1062-1214, utils.py:200-217, test/test_model.py:362-447).  With fit(keep_engine=True) the posteriors
stay on the GPU and the read-out kernels run there (vmr_readout); the same calls on the host copy of rho_f are the check,
and the golden rho_f of the reference's fit pins both."""
import warnings

import numpy as np
import pytest

from tests.golden_util import case_config, load_case

pytestmark = pytest.mark.gpu


def _fit(name, **kw):
    from vimure_amd import VimureModel
    d = load_case(name)
    K, mut, und, seed, priors, fitargs, rho_prior = case_config(d)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VimureModel(mutuality=bool(d["mutuality"]), undirected=und)
        m.fit(d["X"], R=d["R"], K=K, seed=seed, rho_prior=rho_prior, **priors, **fitargs, **kw)
    return d, m


def test_device_readout_equals_host_readout_and_reference_rho():
    d, m = _fit("A_ones_mut", keep_engine=True)
    assert m._rho_f is None and m._engine is not None          # nothing has crossed PCIe yet
    dev = {"rho_max": m.get_inferred_model("rho_max"), "rho_mean": m.get_inferred_model("rho_mean"),
           "fixed": m.get_inferred_model("fixed_threshold", threshold=0.5), "heur": m.get_inferred_model("heuristic_threshold"),
           "predict": m.predict()}
    assert m._rho_f is None
    rho = m.rho_f                                              # now it does
    np.testing.assert_allclose(rho, d["fit_rho_f"], rtol=1e-6, atol=1e-12)
    host = {"rho_max": m.get_inferred_model("rho_max"), "rho_mean": m.get_inferred_model("rho_mean"),
            "fixed": m.get_inferred_model("fixed_threshold", threshold=0.5), "heur": m.get_inferred_model("heuristic_threshold"),
            "predict": m.predict()}
    for k in dev:
        assert dev[k].shape == (m.L, m.N, m.N) and dev[k].dtype == host[k].dtype, k
        if k == "rho_mean":
            np.testing.assert_allclose(dev[k], host[k], rtol=1e-14, atol=1e-300)
        else:
            assert np.array_equal(dev[k], host[k]), k
    # against the reference's own posterior
    ref = d["fit_rho_f"]
    assert np.array_equal(dev["rho_max"], np.argmax(ref, axis=-1))
    thr = 0.54 * m.G_exp_nu - 0.01
    clear = np.abs(ref[..., 1] - thr) > 1e-9
    assert np.array_equal(dev["heur"][clear], (ref[..., 1] >= thr).astype(int)[clear])
    assert dev["rho_max"].sum() > 0
    m.close()
    assert m._engine is None and m.rho_f is rho


def test_readout_errors_warnings_and_sampling():
    d, m = _fit("B_random_mask_K3", keep_engine=True)          # K = 3: threshold methods fall back to rho_max
    with pytest.raises(ValueError, match="'method' should be one of"):
        m.get_inferred_model(method="NotImplemented")
    with pytest.warns(UserWarning, match="threshold methods is incompatible"):
        y = m.get_inferred_model("fixed_threshold", threshold=0.5)
    assert np.array_equal(y, m.get_inferred_model("rho_max")) and y.max() <= 2
    mean = m.get_inferred_model("rho_mean")
    np.testing.assert_allclose(mean, np.dot(d["fit_rho_f"], range(3)), rtol=1e-6, atol=1e-9)
    Y = m.sample_inferred_model(N=4, seed=3)
    assert len(Y) == 4 and all(y.shape == (m.L, m.N, m.N) and y.max() <= 2 for y in Y)
    est = m.get_posterior_estimates()
    assert set(est) == {"nu", "theta", "lambda", "rho"} and est["rho"].shape == (m.L, m.N, m.N, 3)
    np.testing.assert_allclose(est["theta"], d["fit_G_exp_theta_f"], rtol=1e-6)
    m.close()
    d2, m2 = _fit("A_ones_mut")                                # the default fit: rho_f on the host, no engine kept
    assert m2._engine is None
    with pytest.raises(ValueError, match="you must set the threshold"):
        m2.get_inferred_model("fixed_threshold")
    with pytest.raises(ValueError, match="you must set the threshold"):
        m2.get_inferred_model("fixed_threshold", threshold=2)
    d3, m3 = _fit("C_ones_nomut")                              # mutuality off: only rho_max
    with pytest.warns(UserWarning, match="threshold methods is incompatible"):
        y3 = m3.get_inferred_model("rho_mean")
    assert np.array_equal(y3, np.argmax(m3.rho_f, axis=-1))


def test_several_realisations_keep_the_best_on_the_device():
    """num_realisations > 1: the next initial state is drawn on a host thread while the GPU sweeps, the best realisation
    is snapshotted on the device and read back once; the result is the reference's (golden case H: 2 realisations)."""
    d, m = _fit("H_ref_f1_under", keep_engine=True)
    assert int(d["fitarg_num_realisations"]) == 2 and m.trace["realisation"].max() == 1
    assert abs(m.maxL - float(d["fit_maxL"])) <= 1e-8 * abs(float(d["fit_maxL"]))
    np.testing.assert_allclose(m.rho_f, d["fit_rho_f"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(m.gamma_shp_f, d["fit_gamma_shp_f"], rtol=1e-6)
    assert m.seed == int(d["fit_final_seed"])
    m.close()


def test_device_sampler_distribution_and_reproducibility():
    """`sample_inferred_model(device=True)` / vmr_sample: a categorical draw per tie from the rho kept on the GPU (reference
    model.py:1062-1096, synthetic.py:964-1177).  NumPy's PCG64 stream cannot be followed on a GPU, so the device mode is held
    to the DISTRIBUTION: per-tie frequencies over 3000 samples inside binomial bounds of the reference's golden rho_f, both
    for single trials and for the reference's "mode of N trials" form; identical output for identical seeds; the exact host
    mode stays the default."""
    d, m = _fit("B_random_mask_K3", keep_engine=True)          # K = 3, random mask
    eng, ref = m._engine, d["fit_rho_f"]
    L, N, K = m.L, m.N, 3
    a, b = eng.sample(7), eng.sample(7)
    assert a.shape == (L, N, N) and a.dtype == np.uint8 and a.max() <= K - 1
    assert np.array_equal(a, b) and not np.array_equal(a, eng.sample(8))
    S = 3000
    freq = np.zeros((L, N, N, K))
    for s in range(S):
        y = eng.sample(1000 + s)
        for k in range(K):
            freq[..., k] += (y == k)
    freq /= S
    sd = np.sqrt(np.maximum(ref * (1.0 - ref), 1e-12) / S)
    z = np.abs(freq - ref) / (sd + 1.0 / S)
    assert z.max() < 6.5, z.max()                               # ~1.4e5 cells: 6.5 sigma leaves no room for a wrong table
    assert (z > 3.0).mean() < 0.01
    # the reference's sample i of N: argmax of N multinomial trials (ties -> first maximum)
    Y3 = m.sample_inferred_model(N=3, seed=11, device=True)
    assert len(Y3) == 3 and all(y.shape == (L, N, N) for y in Y3)
    assert all(np.array_equal(y, eng.sample(11 + i, n_trials=3)) for i, y in enumerate(Y3))
    hit = np.mean([eng.sample(500 + s, n_trials=3) == np.argmax(ref, axis=-1) for s in range(200)])
    one = np.mean([eng.sample(500 + s, n_trials=1) == np.argmax(ref, axis=-1) for s in range(200)])
    assert hit >= one                                           # the mode of 3 trials sits on argmax rho at least as often
    assert m._rho_f is None                                     # rho never crossed PCIe
    # posterior-predictive network with Y drawn on the device
    from vimure_amd.synthetic import PosteriorSyntheticNetwork
    net = PosteriorSyntheticNetwork(m, seed_Y=7, device=True)
    net.build_Y()
    assert np.array_equal(net.Y.toarray(), a.astype(net.Y.toarray().dtype))
    m.close()
    with pytest.raises(ValueError, match="keep_engine"):
        m.sample_inferred_model(device=True)
