"""ctypes front-end of oracle/cavi_ref.c (CPU oracle, plain C + OpenMP) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libcavi_ref.so")
_dp = C.POINTER(C.c_double)
_u8 = C.POINTER(C.c_uint8)


class _State(C.Structure):
    _fields_ = [("L", C.c_int), ("N", C.c_int), ("M", C.c_int), ("K", C.c_int), ("mut", C.c_int),
                ("X", C.c_void_p), ("R", C.c_void_p), ("eps", C.c_double),
                ("a_th", C.c_void_p), ("b_th", C.c_void_p), ("a_la", C.c_void_p), ("b_la", C.c_void_p),
                ("a_eta", C.c_double), ("b_eta", C.c_double),
                ("gamma_shp", C.c_void_p), ("gamma_rte", C.c_void_p), ("phi_shp", C.c_void_p), ("phi_rte", C.c_void_p),
                ("nu_shp", C.c_double), ("nu_rte", C.c_double),
                ("rho", C.c_void_p), ("logpr", C.c_void_p), ("g_nu_cache", C.c_double)]


def build(force=False):
    src = os.path.join(HERE, "cavi_ref.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(src) > os.path.getmtime(LIB):
        subprocess.run(["make", "-C", HERE, "-s"] + (["-B"] if force else []), check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        for f in ("ref_update_gamma", "ref_update_phi", "ref_update_rho", "ref_update_nu", "ref_cavi_step"):
            getattr(_lib, f).argtypes = [C.POINTER(_State)]
            getattr(_lib, f).restype = None
        _lib.ref_elbo.argtypes = [C.POINTER(_State)]
        _lib.ref_elbo.restype = C.c_double
        _lib.ref_threads.restype = C.c_int
        _lib.ref_set_threads.argtypes = [C.c_int]
    return _lib


class CRef:
    """State holder: arrays are owned here (NumPy), the C code updates them in place."""

    def __init__(self, X, R, K, mutuality, priors, gamma_shp, gamma_rte, phi_shp, phi_rte, nu_shp, nu_rte, pr_rho,
                 eps=1e-12, g_nu_cache=None):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64).copy()
        self.X = np.ascontiguousarray(X, dtype=np.uint8)
        self.R = None if R is None else np.ascontiguousarray(R, dtype=np.uint8)
        L, N, _, M = self.X.shape
        self.a_th, self.b_th = f(np.broadcast_to(priors[0], (L, M))), f(np.broadcast_to(priors[1], (L, M)))
        self.a_la, self.b_la = f(np.broadcast_to(priors[2], (L, K))), f(np.broadcast_to(priors[3], (L, K)))
        self.gamma_shp, self.gamma_rte, self.phi_shp, self.phi_rte = f(gamma_shp), f(gamma_rte), f(phi_shp), f(phi_rte)
        self.rho = f(pr_rho)
        self.logpr = np.log(np.asarray(pr_rho, dtype=np.float64) + eps)
        s = _State()
        s.L, s.N, s.M, s.K, s.mut = L, N, M, int(K), int(bool(mutuality))
        s.X, s.R, s.eps = self.X.ctypes.data, (self.R.ctypes.data if self.R is not None else None), eps
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

    nu_shp = property(lambda self: self.s.nu_shp)

    def update_gamma(self): self.lib.ref_update_gamma(C.byref(self.s))
    def update_phi(self): self.lib.ref_update_phi(C.byref(self.s))
    def update_rho(self): self.lib.ref_update_rho(C.byref(self.s))
    def update_nu(self): self.lib.ref_update_nu(C.byref(self.s))
    def cavi_step(self): self.lib.ref_cavi_step(C.byref(self.s))
    def elbo(self): return self.lib.ref_elbo(C.byref(self.s))
    def threads(self): return self.lib.ref_threads()
    def set_threads(self, n): self.lib.ref_set_threads(int(n))
