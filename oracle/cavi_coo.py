"""ctypes front-end of oracle/cavi_coo.c (CPU oracle over coordinate lists, plain C + OpenMP) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libcavi_coo.so")


class _State(C.Structure):
    _fields_ = [("L", C.c_int), ("N", C.c_int), ("M", C.c_int), ("K", C.c_int), ("mut", C.c_int),
                ("nx", C.c_int64), ("xt", C.c_void_p), ("xm", C.c_void_p), ("xv", C.c_void_p),
                ("r_all", C.c_int), ("nr", C.c_int64), ("rt", C.c_void_p), ("rm", C.c_void_p),
                ("eps", C.c_double),
                ("a_th", C.c_void_p), ("b_th", C.c_void_p), ("a_la", C.c_void_p), ("b_la", C.c_void_p),
                ("a_eta", C.c_double), ("b_eta", C.c_double),
                ("gamma_shp", C.c_void_p), ("gamma_rte", C.c_void_p), ("phi_shp", C.c_void_p), ("phi_rte", C.c_void_p),
                ("nu_shp", C.c_double), ("nu_rte", C.c_double),
                ("rho", C.c_void_p), ("logpr", C.c_void_p), ("g_nu_cache", C.c_double),
                ("xp", C.c_void_p), ("rpn", C.c_void_p), ("xy", C.c_void_p), ("xin", C.c_void_p), ("q", C.c_void_p)]


def build(force=False):
    src = os.path.join(HERE, "cavi_coo.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(src) > os.path.getmtime(LIB):
        subprocess.run(["make", "-C", HERE, "-s"] + (["-B"] if force else []), check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        for f in ("coo_update_gamma", "coo_update_phi", "coo_update_rho", "coo_update_nu", "coo_cavi_step", "coo_release"):
            getattr(_lib, f).argtypes = [C.POINTER(_State)]
            getattr(_lib, f).restype = None
        _lib.coo_prepare.argtypes = [C.POINTER(_State)]
        _lib.coo_prepare.restype = C.c_int
        _lib.coo_elbo.argtypes = [C.POINTER(_State)]
        _lib.coo_elbo.restype = C.c_double
        _lib.coo_threads.restype = C.c_int
    return _lib


def _sorted_lists(subs, shape, vals=None):
    """(l,i,j,m) subscripts -> tie index, reporter (and values) sorted by (tie, reporter); duplicates are an error."""
    L, N, _, M = shape
    l, i, j, m = (np.asarray(a, dtype=np.int64) for a in subs)
    t = (l * N + i) * N + j
    order = np.lexsort((m, t))
    t, m = np.ascontiguousarray(t[order]), np.ascontiguousarray(m[order].astype(np.int32))
    v = None if vals is None else np.ascontiguousarray(np.asarray(vals)[order].astype(np.int32))
    return t, m, v


class CooRef:
    """State holder: arrays are owned here (NumPy), the C code updates them in place.
    X: (subs, vals) with subs a 4-tuple of index arrays (what the reference's sptensor holds), or a dense array.
    R: None (all ones), a 4-tuple of index arrays, or a dense 0/1 array."""

    def __init__(self, X, R, shape, K, mutuality, priors, gamma_shp, gamma_rte, phi_shp, phi_rte, nu_shp, nu_rte, pr_rho,
                 eps=1e-12, g_nu_cache=None):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64).copy()
        L, N, _, M = (int(s) for s in shape)
        if isinstance(X, np.ndarray):
            sx = np.nonzero(X)
            X = (sx, X[sx])
        if isinstance(R, np.ndarray):
            R = np.nonzero(R)
        self.xt, self.xm, self.xv = _sorted_lists(X[0], (L, N, N, M), X[1])
        self.rt = self.rm = None
        if R is not None:
            self.rt, self.rm, _ = _sorted_lists(R, (L, N, N, M))
        self.a_th, self.b_th = f(np.broadcast_to(priors[0], (L, M))), f(np.broadcast_to(priors[1], (L, M)))
        self.a_la, self.b_la = f(np.broadcast_to(priors[2], (L, K))), f(np.broadcast_to(priors[3], (L, K)))
        self.gamma_shp, self.gamma_rte, self.phi_shp, self.phi_rte = f(gamma_shp), f(gamma_rte), f(phi_shp), f(phi_rte)
        self.rho = f(pr_rho)
        assert self.rho.shape == (L, N, N, K)
        self.logpr = np.log(self.rho + eps)
        s = _State()
        s.L, s.N, s.M, s.K, s.mut = L, N, M, int(K), int(bool(mutuality))
        s.nx, s.xt, s.xm, s.xv = len(self.xt), self.xt.ctypes.data, self.xm.ctypes.data, self.xv.ctypes.data
        s.r_all = int(R is None)
        if R is not None:
            s.nr, s.rt, s.rm = len(self.rt), self.rt.ctypes.data, self.rm.ctypes.data
        s.eps = eps
        s.a_th, s.b_th, s.a_la, s.b_la = (a.ctypes.data for a in (self.a_th, self.b_th, self.a_la, self.b_la))
        s.a_eta, s.b_eta = float(priors[4]), float(priors[5])
        s.gamma_shp, s.gamma_rte = self.gamma_shp.ctypes.data, self.gamma_rte.ctypes.data
        s.phi_shp, s.phi_rte = self.phi_shp.ctypes.data, self.phi_rte.ctypes.data
        s.nu_shp, s.nu_rte = float(nu_shp), float(nu_rte)
        s.rho, s.logpr = self.rho.ctypes.data, self.logpr.ctypes.data
        if g_nu_cache is None:
            from scipy.special import psi
            g_nu_cache = float(np.exp(psi(nu_shp) - np.log(nu_rte))) if mutuality else 0.0
        s.g_nu_cache = g_nu_cache
        self.s = s
        self.lib = lib()
        rc = self.lib.coo_prepare(C.byref(self.s))
        if rc != 0:
            raise ValueError("duplicate (l,i,j,m) subscripts in %s" % ("X" if rc == -1 else "R"))

    nu_shp = property(lambda self: self.s.nu_shp)

    def __del__(self):
        try:
            self.lib.coo_release(C.byref(self.s))
        except Exception:
            pass

    def update_gamma(self): self.lib.coo_update_gamma(C.byref(self.s))
    def update_phi(self): self.lib.coo_update_phi(C.byref(self.s))
    def update_rho(self): self.lib.coo_update_rho(C.byref(self.s))
    def update_nu(self): self.lib.coo_update_nu(C.byref(self.s))
    def cavi_step(self): self.lib.coo_cavi_step(C.byref(self.s))
    def elbo(self): return self.lib.coo_elbo(C.byref(self.s))
    def threads(self): return self.lib.coo_threads()
