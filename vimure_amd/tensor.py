"""Minimal COO container with the attribute surface the reference uses from
`sktensor.sptensor` (subs, vals, shape, toarray) -- reference utils.py:115-132, 220-248.
The engine itself works on dense uint8 X and a bit-packed R."""
import numpy as np


class SparseTensor:
    def __init__(self, subs, vals, shape=None, dtype=None):
        if not isinstance(subs, tuple):
            raise ValueError("Subscripts must be a tuple of array-likes")
        self.subs = tuple(np.asarray(s, dtype=np.int64) for s in subs)
        self.vals = np.asarray(vals, dtype=dtype)
        if len(self.subs) and len(self.subs[0]) != len(self.vals):
            raise ValueError("Subscripts and values must be of equal length")
        if shape is None:
            shape = tuple(int(s.max()) + 1 for s in self.subs)
        self.shape = tuple(int(s) for s in shape)
        self.ndim = len(self.shape)
        self.dtype = self.vals.dtype

    def __len__(self):
        return len(self.vals)

    def toarray(self, dtype=None):
        out = np.zeros(self.shape, dtype=dtype or self.dtype)
        if len(self.vals):
            out[self.subs] = self.vals
        return out

    @classmethod
    def fromarray(cls, A):
        A = np.asarray(A)
        subs = np.nonzero(A)
        return cls(subs, A[subs], shape=A.shape, dtype=A.dtype)


def layer_of(X, l):
    """Layer l of a [L,N,N,M] tensor as a one-layer tensor of the same kind (COO container or dense array)."""
    if is_sparse_like(X):
        ls = np.asarray(X.subs[0])
        shape = (1,) + tuple(int(v) for v in X.shape[1:])
        if len(ls) == 0 or bool(np.all(ls[:-1] <= ls[1:])):   # layers in order (np.nonzero order, a sorted edge list): a slice, no copy
            a, b = np.searchsorted(ls, [l, l + 1])
            subs = (np.zeros(int(b - a), np.int64),) + tuple(np.asarray(c)[a:b] for c in X.subs[1:])
            return SparseTensor(subs, np.asarray(X.vals)[a:b], shape=shape)
        keep = ls == l
        subs = (np.zeros(int(keep.sum()), np.int64),) + tuple(np.asarray(a)[keep] for a in X.subs[1:])
        return SparseTensor(subs, np.asarray(X.vals)[keep], shape=shape)
    return np.ascontiguousarray(np.asarray(X)[l:l + 1])


def is_sparse_like(X):
    """Duck-typed COO tensor (ours, or a real sktensor.sptensor if the user has one)."""
    return hasattr(X, "subs") and hasattr(X, "vals") and hasattr(X, "shape")


COUNT_MAX = 2 ** 31 - 1   # counts travel as int32 coordinate values (vmr_create_coo); the reference holds int64 (utils.py:241-242)
M_COO_MAX = 8192          # 13-bit reporter field of the coordinate keys


def engine_data(X, what="X"):
    """The form a count tensor reaches the engine in: a coordinate container (-> vmr_create_coo: any count, M <= 8192) or a dense
    uint8 array (-> vmr_create: counts <= 255, any M).  A dense array with larger counts becomes its coordinate lists."""
    if is_sparse_like(X):
        vals = np.asarray(X.vals)
        if len(vals) and (vals.min() < 0 or vals.max() > COUNT_MAX):
            raise ValueError(f"{what} entries must be integers in [0, 2^31)")
        if int(X.shape[3]) <= M_COO_MAX and (len(vals) == 0 or vals.min() >= 1):
            return X
        if len(vals) and vals.max() > 255:
            raise ValueError(f"{what}: counts above 255 need the coordinate-list layout, which holds M <= {M_COO_MAX} reporters")
        return to_dense_u8(X, what)
    A = np.asarray(X)
    if A.size and A.dtype != np.uint8 and A.max() > 255:
        if A.min() < 0 or A.max() > COUNT_MAX:
            raise ValueError(f"{what} entries must be integers in [0, 2^31)")
        if A.ndim != 4 or A.shape[3] > M_COO_MAX:
            raise ValueError(f"{what}: counts above 255 need the coordinate-list layout, which holds M <= {M_COO_MAX} reporters")
        return SparseTensor.fromarray(A.astype(np.int64) if A.dtype.kind == "f" else A)
    return to_dense_u8(A, what)


def to_dense_u8(X, what="X"):
    """ndarray / COO container -> C-contiguous uint8 [L,N,N,M]; counts must lie in [0,255]."""
    if is_sparse_like(X):
        vals = np.asarray(X.vals)
        if len(vals) and (vals.min() < 0 or vals.max() > 255):
            raise ValueError(f"{what} entries must be integers in [0, 255] for the uint8 device layout")
        out = np.zeros(tuple(int(s) for s in X.shape), np.uint8)
        if len(vals):
            subs = tuple(np.asarray(s, dtype=np.int64) for s in X.subs)
            flat = np.ravel_multi_index(subs, out.shape)
            if len(np.unique(flat)) != len(flat):
                raise ValueError(f"{what} holds repeated (l, i, j, m) subscripts")
            out[subs] = vals.astype(np.uint8)
        return out
    A = np.asarray(X)
    if A.dtype != np.uint8:
        if A.size and (A.min() < 0 or A.max() > 255):
            raise ValueError(f"{what} entries must be integers in [0, 255] for the uint8 device layout")
        A = A.astype(np.int64).astype(np.uint8) if A.dtype.kind == "f" else A.astype(np.uint8)
    return np.ascontiguousarray(A)
