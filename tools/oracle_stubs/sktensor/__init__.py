"""Container-only stand-in for scikit-tensor-py3 0.4.2 (absent offline).

Used ONLY by tools/make_golden.py, in the build container, to import the
read-only reference package and dump golden vectors.  It holds no arithmetic
of the reference: just the COO / dense containers the reference stores its
data in.  It never ships in the product package and never runs on the GPU box.
"""
import numpy as np
from .sptensor import sptensor  # noqa: F401


class dtensor(np.ndarray):
    def __new__(cls, a):
        return np.asarray(a).view(cls)

    def toarray(self):
        return np.asarray(self)
