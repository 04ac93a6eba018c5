"""Object wrapper over the C-ABI: one `CaviEngine` = one dataset on one MI355X.

The engine owns the device copies of X (uint8) and R (bit mask) and the variational
state; the host (`vimure_amd.model.VimureModel`) only draws the RandomState-seeded
initial values and applies the ELBO stop rule.
"""
import ctypes as C

import numpy as np

from . import _lib


class EngineError(RuntimeError):
    pass


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def host_buffer(n_doubles, pinned=True):
    """float64 staging buffer for rho-sized transfers: page-locked (through torch) when that is available and worth it."""
    if pinned and n_doubles * 8 >= (8 << 20):
        try:
            import torch
            if torch.cuda.is_available():
                t = torch.empty(int(n_doubles), dtype=torch.float64, pin_memory=True)
                a = t.numpy()
                return a, t   # (the tensor owns the memory: keep it alive with the array)
        except Exception:
            pass
    return np.empty(int(n_doubles), np.float64), None


class CaviEngine:
    def __init__(self, X, R=None, K=2, mutuality=True, eps=1e-12, device=None):
        self._h = C.c_void_p()
        self._staging = []   # pinned float64 buffers reused across realisations / fits on this engine
        self.lib = _lib.load()
        on_dev = _is_torch(X)
        if on_dev:
            if not X.is_cuda or X.dtype.__str__() != "torch.uint8" or not X.is_contiguous():
                raise ValueError("device X must be a contiguous torch.uint8 tensor on the GPU")
            if R is not None and (not _is_torch(R) or not R.is_cuda or R.dtype.__str__() != "torch.uint8"
                                  or not R.is_contiguous() or tuple(R.shape) != tuple(X.shape)):
                raise ValueError("device R must be a contiguous torch.uint8 GPU tensor shaped like X")
            if device is None:
                device = X.device.index or 0
            import torch
            torch.cuda.synchronize(X.device)
            xp, rp = X.data_ptr(), (R.data_ptr() if R is not None else None)
            shape = tuple(X.shape)
        else:
            if getattr(X, "dtype", np.uint8) != np.uint8 and np.size(X) and (np.min(X) < 0 or np.max(X) > 255):
                raise ValueError("dense X is a uint8 tensor (counts in [0, 255]); larger counts go through CaviEngine.from_coo")
            X = np.ascontiguousarray(X, dtype=np.uint8)
            if R is not None:
                R = np.ascontiguousarray(R, dtype=np.uint8)
                if R.shape != X.shape:
                    raise ValueError("Dimensions of reporter mask (R) do not match L x N x N x M")
            xp, rp = X.ctypes.data, (R.ctypes.data if R is not None else None)
            shape = X.shape
            if device is None:
                device = 0
        if len(shape) != 4 or shape[1] != shape[2]:
            raise ValueError("X must have shape (L, N, N, M)")
        self.L, self.N, _, self.M = (int(s) for s in shape)
        self.K, self.mutuality, self.device = int(K), bool(mutuality), int(device)
        rc = self.lib.vmr_create(C.byref(self._h), self.device, self.L, self.N, self.M, self.K, int(self.mutuality),
                                 xp, rp, int(on_dev), float(eps))
        if rc != 0:
            msg = self.lib.vmr_last_error(None).decode()
            self._h = C.c_void_p()
            raise (ValueError if rc == _lib.VMR_EINVAL else EngineError)(msg)
        self._keep = (X, R)

    @classmethod
    def from_coo(cls, subs, vals, shape, R=None, K=2, mutuality=True, eps=1e-12, device=None):
        """Dataset from coordinate lists -- the reference's own containers (`X.subs`, `X.vals`, `R.subs`; reference
        model.py:136-171) -- without a dense [L,N,N,M] tensor on the host or the device (vmr_create_coo).
        subs: 4 index arrays (l, i, j, m); vals: counts in [1, 2^31); R: None (every reporter may report on every tie) or 4
        index arrays of the mask's non-zeros; NumPy arrays or torch GPU tensors (int32 / int64)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self._staging = []
        self.lib = _lib.load()
        on_dev = _is_torch(vals)
        L, N, N2, M = (int(s) for s in shape)
        if N != N2:
            raise ValueError("X must have shape (L, N, N, M)")

        def cols(arrs):
            if on_dev:
                import torch
                out = [a.to(torch.int32).contiguous() for a in arrs]
                return out, [a.data_ptr() for a in out]
            out = [np.ascontiguousarray(a, dtype=np.int32) for a in arrs]
            return out, [a.ctypes.data for a in out]
        if len(subs) != 4:
            raise ValueError("subs must be the 4 index arrays (l, i, j, m)")
        xk, xp = cols(list(subs) + [vals])
        nx = int(xk[0].shape[0])
        if any(int(a.shape[0]) != nx for a in xk):
            raise ValueError("Subscripts and values must be of equal length")
        if R is None:
            rk, rp, nr = [], [None] * 4, -1
        else:
            if len(R) != 4:
                raise ValueError("R must be the 4 index arrays (l, i, j, m) of the mask's non-zeros")
            rk, rp = cols(list(R))
            nr = int(rk[0].shape[0])
        if device is None:
            device = (vals.device.index or 0) if on_dev else 0
        if on_dev:
            import torch
            torch.cuda.synchronize(vals.device)
        self.L, self.N, self.M = L, N, M
        self.K, self.mutuality, self.device = int(K), bool(mutuality), int(device)
        rc = self.lib.vmr_create_coo(C.byref(self._h), self.device, L, N, M, self.K, int(self.mutuality), nx, *xp, nr, *rp,
                                     int(on_dev), float(eps))
        if rc != 0:
            msg = self.lib.vmr_last_error(None).decode()
            self._h = C.c_void_p()
            raise (ValueError if rc == _lib.VMR_EINVAL else EngineError)(msg)
        self._keep = (xk, rk)
        return self

    # -- helpers
    def _check(self, rc):
        if rc == 0:
            return
        msg = self.lib.vmr_last_error(self._h).decode()
        if rc in (_lib.VMR_EINVAL, _lib.VMR_ENAN):
            raise ValueError(msg)
        raise EngineError(msg)

    def staging(self, i):
        """The i-th rho-sized host staging buffer of this engine (allocated on first use, pinned when possible)."""
        while len(self._staging) <= i:
            self._staging.append(host_buffer(self.L * self.N * self.N * self.K))
        return self._staging[i][0].reshape(self.L, self.N, self.N, self.K)

    def can_upload_ahead(self):
        """Page-locked staging buffers (through torch) exist for this engine's rho size, i.e. `upload_ahead` works."""
        if not hasattr(self, "_can_ahead"):
            self.staging(0)
            self._can_ahead = self._staging[0][1] is not None
        return self._can_ahead

    def upload_ahead(self, i, slot):
        """Copy staging buffer i to a device buffer of this engine (two in rotation: `slot` 0 / 1) on a stream of its own and
        wait for the copy; returns the device tensor (what `set_state` then takes with a device-to-device copy), or None
        when the staging buffer is not page-locked / torch is not there.  For the thread that draws the next realisation
        while the GPU sweeps the current one: the 5 ms upload of a 256 MB pr_rho leaves the critical path."""
        arr, ten = self._staging[i]
        if ten is None:
            return None
        import torch
        if not hasattr(self, "_ahead"):
            self._ahead = [None, None]
            self._ahead_stream = torch.cuda.Stream(device=self.device)
        if self._ahead[slot] is None:
            self._ahead[slot] = torch.empty((self.L, self.N, self.N, self.K), dtype=torch.float64, device=f"cuda:{self.device}")
        with torch.cuda.stream(self._ahead_stream):
            self._ahead[slot].view(-1).copy_(ten, non_blocking=True)
        self._ahead_stream.synchronize()
        return self._ahead[slot]

    def close(self):
        self._ahead = [None, None]
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.vmr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data
    def data_stats(self, coverage=True):
        s = C.c_double()
        cov = np.empty((self.L, self.N, self.N), np.uint8) if coverage else None
        self._check(self.lib.vmr_data_stats(self._h, C.byref(s), cov.ctypes.data if coverage else None))
        return s.value, cov

    # -- parameters
    def set_priors(self, alpha_theta, beta_theta, alpha_lambda, beta_lambda, alpha_eta, beta_eta):
        at = _f64(np.broadcast_to(alpha_theta, (self.L, self.M)))
        bt = _f64(np.broadcast_to(beta_theta, (self.L, self.M)))
        al = _f64(np.broadcast_to(alpha_lambda, (self.L, self.K)))
        bl = _f64(np.broadcast_to(beta_lambda, (self.L, self.K)))
        self._check(self.lib.vmr_set_priors(self._h, at.ctypes.data, bt.ctypes.data, al.ctypes.data, bl.ctypes.data,
                                            float(alpha_eta), float(beta_eta)))

    def set_state(self, gamma_shp, gamma_rte, phi_shp, phi_rte, nu_shp, nu_rte, pr_rho):
        gs, gr, ps, pr = _f64(gamma_shp), _f64(gamma_rte), _f64(phi_shp), _f64(phi_rte)
        assert gs.shape == (self.L, self.M) and gr.shape == gs.shape
        assert ps.shape == (self.L, self.K) and pr.shape == ps.shape
        if _is_torch(pr_rho):
            assert pr_rho.is_cuda and pr_rho.is_contiguous() and tuple(pr_rho.shape) == (self.L, self.N, self.N, self.K)
            import torch
            assert pr_rho.dtype == torch.float64
            if not any(pr_rho is t for t in getattr(self, "_ahead", [])):   # (upload_ahead has waited for its copy already)
                torch.cuda.synchronize(pr_rho.device)
            pp, dev = pr_rho.data_ptr(), 1
        else:
            pr_rho = _f64(pr_rho)
            assert pr_rho.shape == (self.L, self.N, self.N, self.K)
            pp, dev = pr_rho.ctypes.data, 0
        self._check(self.lib.vmr_set_state(self._h, gs.ctypes.data, gr.ctypes.data, ps.ctypes.data, pr.ctypes.data,
                                           float(nu_shp), float(nu_rte), pp, dev))

    # -- CAVI
    def step(self, n_iters=1, want_elbo=False):
        if want_elbo:
            e = C.c_double()
            self._check(self.lib.vmr_step(self._h, int(n_iters), C.byref(e)))
            return e.value
        self._check(self.lib.vmr_step(self._h, int(n_iters), None))
        return None

    def fit_loop(self, max_iter, tol, decision):
        """The realisation's convergence loop on the engine's side (reference model.py:405-426, 1021-1056).
        Returns (trace rows [(iter, elbo, runtime, reached)], last ELBO, iterations, converged)."""
        cap = max_iter // 10 + 2
        n, its, conv, e = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        ri, rr = np.empty(cap, np.int32), np.empty(cap, np.int32)
        re, rt = np.empty(cap), np.empty(cap)
        self._check(self.lib.vmr_fit_loop(self._h, int(max_iter), float(tol), int(decision), cap, C.byref(n), ri.ctypes.data,
                                          re.ctypes.data, rt.ctypes.data, rr.ctypes.data, C.byref(e), C.byref(its), C.byref(conv)))
        k = n.value
        return list(zip(ri[:k].tolist(), re[:k].tolist(), rt[:k].tolist(), [bool(v) for v in rr[:k]])), e.value, its.value, bool(conv.value)

    @staticmethod
    def fit_loop_batch(engines, max_iter, tol, decision):
        """`fit_loop` of several engines in lockstep (vmr_fit_loop_batch): the sweeps of all of them share one launch per kernel.
        Every engine must have had its `set_state`.  Returns one `fit_loop` result per engine, in order."""
        engines = list(engines)
        n = len(engines)
        if n == 0:
            return []
        lib = engines[0].lib
        cap = max_iter // 10 + 2
        hs = (C.c_void_p * n)(*[e._h for e in engines])
        nr, its, conv, rcs = (np.zeros(n, np.int32) for _ in range(4))
        e = np.zeros(n)
        ri, rr = np.empty((n, cap), np.int32), np.empty((n, cap), np.int32)
        re, rt = np.empty((n, cap)), np.empty((n, cap))
        rc = lib.vmr_fit_loop_batch(hs, n, int(max_iter), float(tol), int(decision), cap, nr.ctypes.data, ri.ctypes.data,
                                    re.ctypes.data, rt.ctypes.data, rr.ctypes.data, e.ctypes.data, its.ctypes.data,
                                    conv.ctypes.data, rcs.ctypes.data)
        out = []
        for u, eng in enumerate(engines):
            eng._check(int(rcs[u]))
        if rc != 0:   # (a failure before the per-unit codes were written: argument checks)
            engines[0]._check(int(rc))
        for u, eng in enumerate(engines):
            k = int(nr[u])
            out.append((list(zip(ri[u, :k].tolist(), re[u, :k].tolist(), rt[u, :k].tolist(), [bool(v) for v in rr[u, :k]])),
                        float(e[u]), int(its[u]), bool(conv[u])))
        return out

    def elbo(self):
        e = C.c_double()
        self._check(self.lib.vmr_elbo(self._h, C.byref(e)))
        return e.value

    def sweep_local(self, want_elbo=False):
        """Layer-sharded fits: one sweep on the local layers without committing nu -> (nu_partial, elbo_main, elbo_q)."""
        out = (C.c_double * 3)()
        self._check(self.lib.vmr_sweep_local(self._h, int(want_elbo), out))
        return out[0], out[1], out[2]

    def commit_nu(self, nu_partial_total):
        self._check(self.lib.vmr_commit_nu(self._h, float(nu_partial_total)))

    def stream_ptr(self):
        """The handle's hipStream_t (for collectives queued between the engine's kernels: torch.cuda.ExternalStream)."""
        return int(self.lib.vmr_stream(self._h) or 0)

    def sweep_local_dev(self, out3, want_elbo=False):
        """`sweep_local` that leaves (nu_partial, elbo_main, elbo_q) in the float64 CUDA tensor `out3` (3 elements),
        asynchronously on the engine's stream."""
        assert out3.is_cuda and out3.numel() >= 3 and out3.is_contiguous()
        self._check(self.lib.vmr_sweep_local_dev(self._h, int(want_elbo), out3.data_ptr()))

    def commit_nu_dev(self, total):
        """`commit_nu` from a float64 CUDA tensor holding the summed nu partial in element 0 (no host hop)."""
        self._check(self.lib.vmr_commit_nu_dev(self._h, total.data_ptr()))

    def sample(self, seed, n_trials=1, out=None):
        """One posterior sample of Y [L,N,N] uint8 drawn on the device from the current rho: per tie the most frequent
        category of n_trials categorical trials (`Generator.multinomial(n_trials, rho).argmax(-1)`, reference
        model.py:1062-1096), Philox stream keyed by `seed`.  out: a uint8 CUDA tensor to keep the sample on the device."""
        if out is not None:
            assert out.is_cuda and out.is_contiguous() and out.numel() == self.L * self.N * self.N
            self._check(self.lib.vmr_sample(self._h, int(seed) & (2 ** 64 - 1), int(n_trials), out.data_ptr(), 1))
            return out
        y = np.empty((self.L, self.N, self.N), np.uint8)
        self._check(self.lib.vmr_sample(self._h, int(seed) & (2 ** 64 - 1), int(n_trials), y.ctypes.data, 0))
        return y

    def sub_step(self, which):
        self._check(self.lib.vmr_sub_step(self._h, int(which)))

    def sync(self):
        self._check(self.lib.vmr_sync(self._h))

    def readout(self, method, threshold=0.0):
        """Read-out of the current rho on the device: "rho_max" / "threshold" -> uint8 [L,N,N], "rho_mean" -> float64."""
        code = {"rho_max": _lib.READ_RHO_MAX, "rho_mean": _lib.READ_RHO_MEAN, "threshold": _lib.READ_THRESHOLD}[method]
        out = np.empty((self.L, self.N, self.N), np.float64 if method == "rho_mean" else np.uint8)
        self._check(self.lib.vmr_readout(self._h, code, float(threshold), out.ctypes.data, 0))
        return out

    def snapshot(self):
        """Keep the current posteriors on the device (`_update_optimal_parameters`, reference model.py:925-942)."""
        self._check(self.lib.vmr_snapshot(self._h))

    def restore(self):
        """Make the snapshot the current state again."""
        self._check(self.lib.vmr_restore(self._h))

    def get_state(self, rho=True, rho_out=None):
        """Posteriors as NumPy arrays; rho_out: a C-contiguous float64 buffer to receive rho (e.g. pinned memory)."""
        out = {
            "gamma_shp": np.empty((self.L, self.M)), "gamma_rte": np.empty((self.L, self.M)),
            "phi_shp": np.empty((self.L, self.K)), "phi_rte": np.empty((self.L, self.K)),
        }
        ns, nr = C.c_double(), C.c_double()
        r = None
        if rho:
            r = rho_out if rho_out is not None else np.empty((self.L, self.N, self.N, self.K))
            assert r.dtype == np.float64 and r.flags.c_contiguous and r.size == self.L * self.N * self.N * self.K
            r = r.reshape(self.L, self.N, self.N, self.K)
        self._check(self.lib.vmr_get_state(
            self._h, out["gamma_shp"].ctypes.data, out["gamma_rte"].ctypes.data, out["phi_shp"].ctypes.data,
            out["phi_rte"].ctypes.data, C.addressof(ns), C.addressof(nr), r.ctypes.data if rho else None))
        out["nu_shp"], out["nu_rte"] = ns.value, nr.value
        if rho:
            out["rho"] = r
        return out

    def get_geometric(self):
        gt, gl, gn, gc = np.empty((self.L, self.M)), np.empty((self.L, self.K)), C.c_double(), C.c_double()
        self._check(self.lib.vmr_get_geometric(self._h, gt.ctypes.data, gl.ctypes.data, C.addressof(gn),
                                               C.addressof(gc)))
        return gt, gl, gn.value, gc.value

    def data_format(self):
        """("sparse" | "dense", non-zero counts in X): the layout vmr_create chose for this dataset."""
        sp, nnz = C.c_int(), C.c_uint64()
        self._check(self.lib.vmr_data_format(self._h, C.byref(sp), C.byref(nnz)))
        return ("sparse" if sp.value else "dense"), int(nnz.value)

    def mask_format(self):
        """("lists" | "words", listed reporters): whether partial mask rows are also held as short reporter lists."""
        li, n = C.c_int(), C.c_uint64()
        self._check(self.lib.vmr_mask_format(self._h, C.byref(li), C.byref(n)))
        return ("lists" if li.value else "words"), int(n.value)

    def sweep_shape(self):
        """(passes over the entries per sweep, mirror-count levels of the statistics kept in LDS, reports in the far lists): the
        shape vmr_create chose for a sweep of this dataset (vmr_sweep_shape)."""
        p, lv, n = C.c_int(), C.c_int(), C.c_uint64()
        self._check(self.lib.vmr_sweep_shape(self._h, C.byref(p), C.byref(lv), C.byref(n)))
        return int(p.value), int(lv.value), int(n.value)

    # -- measurement
    def profile(self, enable=True):
        """HIP events around the engine's kernels; enable=2: only around the passes over the data (not the finalize kernels)."""
        self._check(self.lib.vmr_profile(self._h, int(enable)))

    def profile_read(self):
        out = {}
        for i, name in enumerate(_lib.KERNEL_NAMES):
            ms, n, b = C.c_double(), C.c_int64(), C.c_double()
            self._check(self.lib.vmr_profile_read(self._h, i, C.byref(ms), C.byref(n)))
            self._check(self.lib.vmr_kernel_bytes(self._h, i, C.byref(b)))
            out[name] = {"ms": ms.value, "launches": n.value, "bytes_per_launch": b.value}
        return out
