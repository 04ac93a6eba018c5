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
