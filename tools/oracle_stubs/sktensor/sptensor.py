import numpy as np


class sptensor:
    """COO container: tuple of index arrays + values (order preserved)."""

    def __init__(self, subs, vals, shape=None, dtype=None, accumfun=None, issorted=False):
        if not isinstance(subs, tuple):
            raise ValueError("Subscripts must be a tuple of array-likes")
        if len(subs) and len(subs[0]) != len(vals):
            raise ValueError("Subscripts and values must be of equal length")
        self.subs = tuple(np.asarray(s) for s in subs)
        self.vals = np.array(vals, dtype=dtype)
        if dtype is None:
            dtype = self.vals.dtype
        self.dtype = dtype
        if shape is None:
            shape = tuple(int(np.max(s)) + 1 for s in self.subs)
        self.shape = tuple(int(s) for s in shape)
        self.ndim = len(self.subs)

    def __len__(self):
        return len(self.vals)

    def __getitem__(self, idx):
        idx = tuple(idx) if isinstance(idx, (tuple, list)) else (idx,)
        hit = np.ones(len(self.vals), dtype=bool)
        for s, i in zip(self.subs, idx):
            hit &= (s == i)
        w = np.nonzero(hit)[0]
        if len(w) == 0:
            return 0
        return self.vals[w[:1]]

    def toarray(self):
        out = np.zeros(self.shape, dtype=self.dtype)
        if len(self.vals):
            out.put(np.ravel_multi_index(self.subs, self.shape), self.vals)
        return out


def fromarray(A):
    A = np.asarray(A)
    subs = np.nonzero(A)
    return sptensor(subs, A[subs], shape=A.shape, dtype=A.dtype)
