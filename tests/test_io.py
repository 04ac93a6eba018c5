"""Edgelist ingestion (vimure_amd/_io.py) against vectors dumped from the reference's `read_from_edgelist`
(tests/golden/J_edgelist_io.npz, made by tools/make_golden.py) and its error/warning behaviour
(reference _io.py:364-511, test/test_io.py)."""
import os
import warnings

import numpy as np
import pandas as pd
import pytest

from tests.golden_util import GOLDEN, load_case
from vimure_amd._io import read_from_edgelist, read_from_igraph, self_reporter_coo

J = dict(np.load(os.path.join(GOLDEN, "J_edgelist_io.npz"), allow_pickle=False))


def _df(prefix="df_"):
    cols = {k[len(prefix):]: J[k] for k in J if k.startswith(prefix)}
    order = [c for c in ("reporter", "ego", "alter", "weight", "layer") if c in cols]
    df = pd.DataFrame({c: cols[c] for c in order})
    for c in ("reporter", "ego", "alter", "layer"):
        df[c] = df[c].astype(str) if df[c].dtype.kind in "US" else df[c]
    return df


def _dense(subs, vals, shape):
    out = np.zeros(shape, np.int64)
    np.add.at(out, tuple(np.asarray(s, dtype=np.int64) for s in subs), vals)
    return out


@pytest.mark.parametrize("tag,kw", [
    ("plain", {}), ("weighted", dict(is_weighted=True)), ("undirected", dict(is_undirected=True, is_weighted=True)),
    ("lists", dict(nodes=sorted(f"n{i:02d}" for i in range(14)), reporters=sorted(f"n{i:02d}" for i in range(10)), K=3)),
])
def test_matches_reference_reader(tag, kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = read_from_edgelist(_df(), **kw)
    assert [net.L, net.N, net.M, net.K] == J[f"{tag}_LNMK"].tolist()
    assert net.nodeNames["name"].tolist() == J[f"{tag}_nodes"].tolist()
    assert list(net.layerNames) == J[f"{tag}_layers"].tolist()
    shape = tuple(J[f"{tag}_shape"])
    assert tuple(net.X.shape) == shape and tuple(net.R.shape) == shape
    want_x = _dense(J[f"{tag}_X_subs"], J[f"{tag}_X_vals"], shape)
    assert np.array_equal(net.X.toarray(np.int64), want_x)
    want_r = _dense(J[f"{tag}_R_subs"], 1, shape) > 0
    assert np.array_equal(net.R.toarray(np.int64) > 0, want_r)


def test_karnataka_village1_money():
    """Same X and R as the reference builds from the village-1 'money' edgelist (its test data:
    reference test/__init__.py:30-41); expected tensors are the COO inputs of golden case I."""
    d = load_case("I_karnataka_vil1_money")
    df = _df("vil1_df_")
    if "weight" in df:
        df["weight"] = df["weight"].astype(float)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = read_from_edgelist(df, K=2)
    assert [net.L, net.N, net.M, net.K] == J["vil1_LNMK"].tolist() == [1, 324, 324, 2]
    assert [str(x) for x in net.nodeNames["name"].tolist()] == [str(x) for x in J["vil1_nodes"].tolist()]
    assert np.array_equal(net.X.toarray(np.uint8), d["X"])
    assert np.array_equal((net.R.toarray(np.uint8) > 0).astype(np.uint8), d["R"])


def test_self_reporter_mask_counts():
    subs = self_reporter_coo(2, 7, [0, 3, 6])
    assert len(subs[0]) == 2 * 3 * 2 * 6                     # L * reporters * (2N - 2)
    R = np.zeros((2, 7, 7, 7), int)
    R[subs] = 1
    assert R[0, 3, 5, 3] == 1 and R[0, 5, 3, 3] == 1 and R[0, 3, 3, 3] == 0 and R[0, 1, 2, 3] == 0 and R[:, :, :, 1].sum() == 0


def test_errors_and_warnings():
    df = _df()
    with pytest.raises(ValueError, match="'df' should be a DataFrame"):
        read_from_edgelist(df.values)
    with pytest.raises(ValueError, match="Required columns not found in data frame: alter"):
        read_from_edgelist(df.drop(columns=["alter"]))
    with pytest.raises(ValueError, match="'nodes' should be a list"):
        read_from_edgelist(df, nodes=("a",))
    with pytest.raises(ValueError, match="does not contain all nodes"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            read_from_edgelist(df, nodes=["n00", "n01"])
    with pytest.raises(ValueError, match="Some reporters in the data frame do not appear"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            read_from_edgelist(df, reporters=["n00"])
    bad = df.copy()
    bad.loc[0, "reporter"] = "outsider"
    with pytest.raises(ValueError, match="some reporters are not nodes"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            read_from_edgelist(bad)
    with pytest.warns(UserWarning) as rec:
        read_from_edgelist(df.drop(columns=["layer", "weight"]))
    msgs = " | ".join(str(w.message) for w in rec)
    for piece in ("The set of nodes was not informed", "The set of reporters was not informed",
                  "Reporters Mask was not informed", "Parameter K was None. Defaulting to: 2"):
        assert piece in msgs
    with pytest.raises(ValueError, match="Dimensions of reporter mask"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            read_from_edgelist(df, R=np.ones((2, 3, 3, 3)))


def test_igraph_like_input():
    class ES:
        def __init__(self, rows): self.rows = rows
        def attributes(self): return ["reporter", "layer", "weight"]
        def __getitem__(self, i): return self.rows[i]

    class G:
        def __init__(self):
            self.vs = {"name": ["a", "b", "c"]}
            self._e = [(0, 1), (1, 2), (2, 0)]
            self.es = ES([{"reporter": "a", "layer": "x", "weight": 1}, {"reporter": "b", "layer": "x", "weight": 1},
                          {"reporter": "c", "layer": "x", "weight": 1}])
        def get_edgelist(self): return self._e
        def ecount(self): return len(self._e)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = read_from_igraph(G())
    X = net.X.toarray(int)
    assert net.N == 3 and X[0, 0, 1, 0] == 1 and X[0, 1, 2, 1] == 1 and X[0, 2, 0, 2] == 1 and X.sum() == 3


def test_undirected_repeated_rows_sum_before_the_max_with_the_transpose():
    """A repeated (layer, ego, alter, reporter) row of different weight survives drop_duplicates; the reference puts the rows
    in a scipy COO matrix, whose duplicates ADD UP when `sparse_max(X, X.T)` converts it (utils.py:184-192, _io.py:276-285)."""
    import pandas as pd
    from scipy import sparse
    from vimure_amd._io import read_from_edgelist
    df = pd.DataFrame({"ego": ["a", "a", "b", "c"], "alter": ["b", "b", "a", "a"], "reporter": ["a", "a", "a", "c"],
                       "weight": [1, 2, 1, 4], "layer": ["x"] * 4})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = read_from_edgelist(df, is_undirected=True, is_weighted=True)
    X = net.X.toarray()
    ids = dict(zip(net.nodeNames["name"], net.nodeNames["id"]))
    a, b, c = ids["a"], ids["b"], ids["c"]
    # what scipy does for reporter a: COO with the repeated entry, max with its transpose
    A = sparse.coo_matrix(([1, 2, 1], ([a, a, b], [b, b, a])), shape=(3, 3))
    gt = (A > A.T).astype(int)
    ref = (gt.multiply(A - A.T) + A.T).toarray()
    assert ref[a, b] == 3 and ref[b, a] == 3
    assert np.array_equal(X[0, :, :, a], ref)
    assert X[0, c, a, c] == 4 and X[0, a, c, c] == 4
