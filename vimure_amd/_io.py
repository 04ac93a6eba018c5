"""Edgelist ingestion: DataFrame / CSV / igraph-like -> observed tensor X and reporter mask R.

Mirrors the reference's readers (src/python/vimure/_io.py:132-360, checks :364-511): same
arguments, defaults, warnings and error messages, same node / layer ordering, the same
"reporters report on their own ties" mask when R is not given (_io.py:213-242), and the same
quirk that the reporter dimension of the tensors is N, not len(reporters) (_io.py:230, 253).
X and R come back as COO `SparseTensor`s (the engine packs them to its dense-u8 / bit layout).
"""
import warnings

import numpy as np
import pandas as pd

from ._log import setup_logging
from .tensor import SparseTensor

module_logger = setup_logging("vm.io", False)


class RealNetwork:
    """Observed network (reference _io.py:93-127)."""

    def __init__(self, X, R, L, N, M, K, nodeNames=None, layerNames=None, seed=10, **kwargs):
        dim = (L, N, N, N)
        if tuple(X.shape) != dim:
            raise ValueError("X has to be a tensor of %s dimensions!" % str(dim))
        self.X, self.R = X, R
        self.L, self.N, self.M, self.K = L, N, M, K
        self.seed = seed
        self.prng = np.random.RandomState(seed)
        if nodeNames is not None:
            self.nodeNames = pd.DataFrame(nodeNames.items(), columns=["id", "name"])
        if layerNames is not None:
            self.layerNames = layerNames

    def getX(self):
        return self.X

    def __repr__(self):
        return f"{self.__class__.__name__} (N={self.N}, M={self.M}, L={self.L}, K={self.K}, seed={self.seed})"


def _check_params_consistency(df, nodes, reporters, ego, alter, reporter, layer, weight):
    """reference _io.py:364-473"""
    if not isinstance(df, pd.DataFrame):
        raise ValueError(f"'df' should be a DataFrame, instead it is of type: {type(df)}.")
    missing = [c for c in (ego, alter, reporter) if c not in df.columns]
    if missing:
        raise ValueError(
            f"Required columns not found in data frame: {', '.join(missing)}. "
            "Mapping used: "
            f"ego='{ego}', alter='{alter}', reporter='{reporter}'. "
            "Hint: Use params ego,alter,... for mapping column names.")
    if nodes is not None and not isinstance(nodes, list):
        raise ValueError(f"'nodes' should be a list, instead it is of type: {type(nodes)}.")
    if reporters is not None and not isinstance(reporters, list):
        raise ValueError(f"'reporters' should be a list, instead it is of type: {type(reporters)}.")
    if nodes == []:
        warnings.warn("The set of nodes was not informed, "
                      f"using {ego} and {alter} columns to infer nodes.", UserWarning)
        nodes = pd.concat([df[ego], df[alter]]).unique().tolist()
    if np.logical_or(~df[ego].isin(nodes), ~df[alter].isin(nodes)).any():
        raise ValueError("A list of nodes was informed, "
                         "but it does not contain all nodes in the data frame.")
    if layer not in df.columns:
        df.loc[:, layer] = "1"
    if weight not in df.columns:
        df.loc[:, weight] = 1
    not_nodes = ("This survey setup is not currently supported by the package: "
                 " some reporters are not nodes in the network. "
                 "Hint: If this is unexpected behaviour, "
                 f"compare the unique values of the `{str(reporter)}` column "
                 f"with those of the `{str(ego)}` and `{str(alter)}` columns.")
    in_df = df[reporter].unique().tolist()
    if reporters is None or reporters == []:
        warnings.warn("The set of reporters was not informed, "
                      "assuming set(reporters) = set(nodes) and N = M.", UserWarning)
        reporters = nodes[:]
        if not set(in_df).issubset(reporters):
            raise ValueError(not_nodes)
    elif not set(in_df).issubset(reporters):
        raise ValueError("Some reporters in the data frame do not appear "
                         "in the list of reporters provided. "
                         f"Hint: Compare the unique values of the `{str(reporter)}` column "
                         "with the list of reporters passed as parameter.")
    if not set(reporters).issubset(nodes):
        raise ValueError(not_nodes)
    if not set(nodes).issubset(reporters):
        warnings.warn("Not necessarily a problem, but"
                      " some of the nodes are not reporters.", UserWarning)
    return df, nodes, reporters


def self_reporter_coo(L, N, reporter_ids):
    """COO subscripts of R[l,i,j,m] = 1 iff m is a reporter and m in {i, j}, i != j (reference _io.py:230-242)."""
    rep = np.asarray(sorted(reporter_ids), dtype=np.int64)
    others = np.arange(N, dtype=np.int64)
    subs = [[], [], [], []]
    for l in range(L):
        for r in rep:
            o = others[others != r]
            # reference order inside a reporter's N x N matrix: np.nonzero (row-major) of max(A, A^T)
            i = np.concatenate([np.full(N - 1, r), o])
            j = np.concatenate([o, np.full(N - 1, r)])
            order = np.lexsort((j, i))
            subs[0].append(np.full(2 * (N - 1), l)); subs[1].append(i[order]); subs[2].append(j[order])
            subs[3].append(np.full(2 * (N - 1), r))
    if not subs[0]:
        return tuple(np.zeros(0, np.int64) for _ in range(4))
    return tuple(np.concatenate(s) for s in subs)


def read_from_edgelist(df, nodes: list = [], reporters: list = [], is_weighted: bool = False,
                       is_undirected: bool = False, reporter: str = "reporter", layer: str = "layer", ego: str = "ego",
                       alter: str = "alter", weight: str = "weight", K=None, R=None, **kwargs):
    """Edgelist -> RealNetwork (reference _io.py:132-295)."""
    df, nodes, reporters = _check_params_consistency(df, nodes, reporters, ego, alter, reporter, layer, weight)
    layers = sorted(df[layer].unique())
    L, N, M = len(layers), len(nodes), len(reporters)
    df = df[[ego, alter, reporter, layer, weight]].drop_duplicates()
    node_id = {name: i for i, name in enumerate(nodes)}
    layer_id = {name: i for i, name in enumerate(layers)}

    if R is None:
        warnings.warn("Reporters Mask was not informed (parameter R). "
                      "Parser will build it from reporter column, "
                      "assuming a reporter can only report their own ties.", UserWarning)
        rs = self_reporter_coo(L, N, [node_id[r] for r in reporters])
        R = SparseTensor(rs, np.ones(len(rs[0])), shape=(L, N, N, N))
    elif tuple(R.shape) != (L, N, N, M):
        msg = "Dimensions of reporter mask (R) do not match L x N x N x M"
        module_logger.error(msg)
        raise ValueError(msg)

    li = df[layer].map(layer_id).values.astype(np.int64)
    ei = df[ego].map(node_id).values.astype(np.int64)
    ai = df[alter].map(node_id).values.astype(np.int64)
    ri = df[reporter].map(node_id).values.astype(np.int64)
    w = df[weight].values
    data = w if is_weighted else (w > 0).astype("int")
    keep = data > 0
    li, ei, ai, ri, data = li[keep], ei[keep], ai[keep], ri[keep], np.asarray(data)[keep]
    dense_key = ((li * N + ei) * N + ai) * N + ri
    # repeated (l, ego, alter, reporter) rows add up, as the scipy COO matrix the reference builds does (_io.py:276-281)
    order = np.argsort(dense_key, kind="stable")
    dense_key, data = dense_key[order], np.asarray(data)[order]
    uniq, start = np.unique(dense_key, return_index=True)
    data = np.add.reduceat(data, start) if len(start) else data
    dense_key = uniq
    if is_undirected:   # then the element-wise max with the transpose, per (reporter, layer) (_io.py:283-285, utils.sparse_max)
        l_, e_, a_, r_ = np.unravel_index(dense_key, (L, N, N, N)) if len(dense_key) else (np.zeros(0, np.int64),) * 4
        key_t = ((l_ * N + a_) * N + e_) * N + r_
        dense_key = np.concatenate([dense_key, key_t])
        data = np.concatenate([data, data])
        order = np.argsort(dense_key, kind="stable")
        dense_key, data = dense_key[order], data[order]
        uniq, start = np.unique(dense_key, return_index=True)
        data = np.maximum.reduceat(data, start) if len(start) else data
        dense_key = uniq
    subs = np.unravel_index(dense_key, (L, N, N, N)) if len(dense_key) else tuple(np.zeros(0, np.int64) for _ in range(4))
    X = SparseTensor(tuple(np.asarray(s, dtype=np.int64) for s in subs), np.asarray(data), shape=(L, N, N, N))

    if K is None:
        K = int(np.max(X.vals)) + 1
        warnings.warn(f"Parameter K was None. Defaulting to: {K}", UserWarning)
    return RealNetwork(X=X, R=R, L=L, N=N, M=M, K=K, nodeNames={i: n for i, n in enumerate(nodes)}, layerNames=layers,
                       **kwargs)


def read_from_csv(filename: str, **kwargs):
    """reference _io.py:297-321"""
    return read_from_edgelist(pd.read_csv(filename), **kwargs)


def read_from_igraph(G, **kwargs):
    """igraph.Graph (duck-typed: get_edgelist, es, vs) -> RealNetwork (reference _io.py:323-356)."""
    edgelist = G.get_edgelist()
    attrs = list(G.es.attributes())
    df = pd.DataFrame(edgelist, columns=["source", "target"])
    for a in attrs:
        df[a] = [G.es[i][a] for i in range(G.ecount())]
    names = G.vs["name"]
    df = df.rename(columns={"source": "ego", "target": "alter"})
    df["ego"] = [names[i] for i in df["ego"]]
    df["alter"] = [names[i] for i in df["alter"]]
    return read_from_edgelist(df, **kwargs)


def parse_graph_from_networkx(G, **kwargs):
    import networkx as nx
    return read_from_edgelist(nx.to_pandas_edgelist(G), **kwargs)
