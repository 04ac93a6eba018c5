"""GPU: the batch driver (engine reused across seeds) gives the same fits as separate `VimureModel.fit` calls."""
import warnings

import numpy as np
import pytest

from tests.golden_util import load_case

pytestmark = pytest.mark.gpu


def test_fit_layers_reuses_engine_and_matches_single_fits():
    from vimure_amd import VimureModel
    from vimure_amd.batch import fit_layers, fit_datasets
    d = load_case("D_self_mask")
    df = fit_layers(d["X"], d["R"], K=2, seeds=[1, 2, 3], num_realisations=2, max_iter=21, keep_posteriors=True)
    assert len(df) == 2 * 3 and set(df["layer"]) == {0, 1}
    for l in range(2):
        for seed in (1, 2, 3):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = VimureModel().fit(d["X"][l:l + 1], R=d["R"][l:l + 1], K=2, seed=seed, num_realisations=2, max_iter=21)
            got = float(df[(df["layer"] == l) & (df["seed"] == seed)]["elbo"].iloc[0])
            assert abs(got - m.maxL) <= 1e-9 * max(1.0, abs(m.maxL))
    best = df.attrs["best"]
    assert best[0].maxL == pytest.approx(df[df["layer"] == 0]["elbo"].max(), rel=1e-12)
    allrows = fit_datasets({"a": (d["X"], d["R"]), "b": (d["X"][:1], d["R"][:1])}, K=2, seeds=[1, 2],
                           num_realisations=1, max_iter=11)
    assert len(allrows) == (2 + 1) * 2 and set(allrows["dataset"]) == {"a", "b"}
    a0 = allrows[(allrows["dataset"] == "a") & (allrows["layer"] == 0)]["elbo"].values
    b0 = allrows[(allrows["dataset"] == "b") & (allrows["layer"] == 0)]["elbo"].values
    np.testing.assert_allclose(a0, b0, rtol=1e-9)


def test_concurrent_units_equal_serial_units_and_golden():
    """Units run on several host threads / HIP streams at once give, fit by fit, what they give one after the other,
    and the reference's Karnataka village-1 fit (golden case I) comes out of the concurrent driver unchanged."""
    from vimure_amd.batch import fit_datasets, run_karnataka
    d = load_case("D_self_mask")
    k = load_case("I_karnataka_vil1_money")
    data = {"a": (d["X"], d["R"]), "b": (d["X"][:1], d["R"][:1]), "c": (d["X"][1:], d["R"][1:])}
    kw = dict(K=2, seeds=[1, 2, 3], num_realisations=2, max_iter=21)
    ser = fit_datasets(data, workers=1, **kw)
    par = fit_datasets(data, workers=4, **kw)
    assert len(par) == len(ser) == 4 * 3
    assert par[["dataset", "layer", "seed", "iters"]].equals(ser[["dataset", "layer", "seed", "iters"]])
    np.testing.assert_allclose(par["elbo"].values, ser["elbo"].values, rtol=1e-9)
    # ... and in two worker processes beside each other on this GPU (coordinate containers travel to them as they are)
    from vimure_amd.batch import shutdown_pools
    from vimure_amd.tensor import SparseTensor
    coo = {n: (SparseTensor.fromarray(X), SparseTensor.fromarray(R)) for n, (X, R) in data.items()}
    try:
        prc = fit_datasets(coo, workers=2, processes=2, **kw)
    finally:
        shutdown_pools()
    assert prc[["dataset", "layer", "seed", "iters"]].equals(ser[["dataset", "layer", "seed", "iters"]])
    np.testing.assert_allclose(prc["elbo"].values, ser["elbo"].values, rtol=1e-9)
    # karnataka.main over "villages": the four tables; village 1 'money' is the reference's own fit
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        villages = {"vil1": (k["X"], k["R"], ["money"]), "toy": (d["X"], d["R"], ["l0", "l1"])}
        tabs = run_karnataka(villages, seeds=[1], out_dir=tmp, num_realisations=1, max_iter=21, workers=3)
        s = tabs["summary"]
        assert len(s) == 3 and set(s["village"]) == {"vil1", "toy"}
        row = s[(s["village"] == "vil1")].iloc[0]
        assert abs(row["best_elbo"] - float(k["fit_maxL"])) <= 1e-8 * abs(float(k["fit_maxL"]))
        assert row["best_seed"] == int(k["fit_final_seed"]) and row["initial_seed"] == 1
        e = tabs["edgelist"]
        ev = e[e["village"] == "vil1"]
        ref_rho = k["fit_rho_f"][0, ev["source"].values, ev["target"].values, 1]
        np.testing.assert_allclose(ev["vimure_posterior_probability"].values, ref_rho, rtol=1e-6, atol=1e-12)
        assert (ev["in_union"] | ev["in_intersection"] | ev["in_vimure"]).all()
        rel = tabs["reliability"]
        np.testing.assert_allclose(rel[rel["village"] == "vil1"]["theta"].values, k["fit_G_exp_theta_f"][0], rtol=1e-6)
        import os
        import pandas as pd
        assert sorted(os.listdir(tmp)) == sorted(["vimure_model_summary.csv", "vimure_model_trace.csv",
                                                  "vimure_model_edgelist.csv", "vimure_model_reliability.csv"])
        assert len(pd.read_csv(os.path.join(tmp, "vimure_model_summary.csv"))) == 3
        again = run_karnataka(villages, seeds=[1], out_dir=tmp, num_realisations=1, max_iter=21, workers=3)   # resume: all present
        assert len(again["summary"]) == 0 and len(pd.read_csv(os.path.join(tmp, "vimure_model_summary.csv"))) == 3
