"""Host side of the estimator: the reference's `VimureModel` API over the HIP engine.

Mirrors latentnetworks/vimure `src/python/vimure/model.py`: constructor (:39-71), `fit`
(:327-448) with the same keyword set, warnings and error messages (:79-325), the
RandomState draw order of `_set_rho_prior` / `_initialize_priors` (:458-605) so fixed-seed
fits start from the reference's state, the ELBO stop rule (:1021-1056), best-realisation
selection (:428-437, :925-942) and the read-out methods (:1062-1214).

What does NOT happen here: any CAVI arithmetic.  The sweeps and the ELBO run in
libvimure_hip.so (`vimure_amd.engine.CaviEngine`); without it `fit` raises.
"""
import time
import warnings

import numpy as np
import pandas as pd
import scipy.special as sp
from scipy.stats import poisson

from ._log import setup_logging
from .engine import CaviEngine, host_buffer
from .tensor import SparseTensor, engine_data, is_sparse_like, to_dense_u8

try:  # the reference is an sklearn estimator (model.py:28); keep that surface when sklearn is there
    from sklearn.base import BaseEstimator, TransformerMixin
except Exception:  # pragma: no cover
    class BaseEstimator:  # type: ignore
        pass

    class TransformerMixin:  # type: ignore
        pass

INF = 1e10
DEFAULT_EPS = 1e-12
DEFAULT_BIAS0 = 0.0
DEFAULT_MAX_ITER = 500
DEFAULT_NUM_REALISATIONS = 1

_EXTRA = ["R", "EPS", "K", "bias0", "max_iter", "alpha_lambda", "beta_lambda", "alpha_teta", "beta_teta",
          "num_realisations"]  # the reference's whitelist, typos included (model.py:90-101)
_OURS = ["device", "alpha_theta", "beta_theta", "engine", "keep_engine"]


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class VimureModel(TransformerMixin, BaseEstimator):
    """ViMuRe: latent network Y (rho), reporter reliabilities (theta), tie-strength rates
    (lambda) and mutuality (eta), by coordinate-ascent variational inference on an MI355X."""

    def __init__(self, undirected: bool = False, mutuality: bool = True, convergence_tol: float = 0.1,
                 decision: int = 1, verbose: bool = False):
        self.undirected = undirected
        if undirected:
            warnings.warn("Overriding mutuality to False because the network is undirected")
            self.mutuality = False
        else:
            self.mutuality = mutuality
        self.convergence_tol = convergence_tol
        self.decision = decision
        self.verbose = verbose
        self.logger = setup_logging("vm.model.VimureModel", verbose)

    # ------------------------------------------------------------------ parameters (model.py:79-325)
    def _check_fit_params(self, X, lambda_prior, theta_prior, eta_prior, rho_prior, seed, **extra):
        for p in extra:
            if p not in _EXTRA and p not in _OURS:
                self.logger.warning("Ignoring unrecognised parameter %s." % p)

        if isinstance(X, pd.DataFrame) or type(X).__name__ == "Graph":
            from ._io import read_from_edgelist, read_from_igraph
            net = read_from_edgelist(X) if isinstance(X, pd.DataFrame) else read_from_igraph(X)
            X = net.X
            self.nodeNames, self.layerNames = net.nodeNames, net.layerNames
            self.R = net.R
            if extra.get("K") is None:
                self.K = net.K

        dev_tensor = _is_torch(X)
        # a coordinate container (the reference's sptensor surface: subs / vals / shape) goes to the device as it is
        # (vmr_create_coo) when the report lists can hold it; no dense [L,N,N,M] array is built then
        # (and so does a dense array whose counts pass 255: the engine takes any count the reference's int64 holds, below 2^31)
        coo = False
        if dev_tensor or (extra.get("engine") is not None and is_sparse_like(X)):
            Xd = X   # (with `engine` the data is on the device already: only the shape is needed)
            shape = tuple(int(s) for s in X.shape)
        else:
            Xd = engine_data(X, "X") if extra.get("engine") is None else to_dense_u8(X, "X")
            if self.undirected and is_sparse_like(Xd) and (len(Xd.vals) == 0 or np.max(Xd.vals) <= 255):
                Xd = to_dense_u8(Xd, "X")   # (the symmetry check below reads the array)
            coo = is_sparse_like(Xd)
            shape = tuple(int(s) for s in Xd.shape)
        if len(shape) != 4 or shape[1] != shape[2]:
            raise ValueError("X must have shape (L, N, N, M)")
        self.L, self.N, self.M = shape[0], shape[1], shape[3]

        if self.undirected:
            if coo:   # counts beyond a byte: compare the coordinate lists with their (i, j)-swapped selves
                sl, si, sj, sm = (np.asarray(a, dtype=np.int64) for a in Xd.subs)
                v = np.asarray(Xd.vals)
                ka = np.ravel_multi_index((sl, si, sj, sm), shape)
                kb = np.ravel_multi_index((sl, sj, si, sm), shape)
                oa, ob = np.argsort(ka), np.argsort(kb)
                sym = bool(np.array_equal(ka[oa], kb[ob]) and np.array_equal(v[oa], v[ob]))
            else:
                sym = bool((Xd == Xd.transpose(1, 2)).all()) if dev_tensor else np.array_equal(Xd, Xd.transpose(0, 2, 1, 3))
            if not sym:
                msg = "If undirected is True, the given network has to be symmetric wrt l and m!"
                self.logger.error(msg)
                raise ValueError(msg)

        if not hasattr(self, "K"):
            if extra.get("K") is not None:
                self.K = int(extra["K"])
            else:
                self.K = (int(np.max(Xd.vals)) if is_sparse_like(Xd) else int(Xd.max())) + 1
                warnings.warn(f"Parameter K was None. Defaulting to: {self.K}", UserWarning)

        if not hasattr(self, "R"):
            if "R" in extra and extra["R"] is not None:
                R = extra["R"]
                if tuple(int(s) for s in R.shape) != (self.L, self.N, self.N, self.M):
                    msg = "Dimensions of reporter mask (R) do not match L x N x N x M"
                    self.logger.error(msg)
                    raise ValueError(msg)
                self.R = R
            else:
                msg = "Reporters Mask was not informed (parameter R). "
                msg += "The model will assume that every reporter can report on any tie."
                warnings.warn(msg, UserWarning)
                self.R = None  # the engine treats NULL as all ones; no [L,N,N,M] float64 array is built
        Rd = self.R
        if extra.get("engine") is not None:
            Rd = None   # the engine already holds X and R on the device: no pass over the host copies
        elif coo and (Rd is None or is_sparse_like(Rd)):
            pass        # coordinate lists: handed to the engine as they are
        elif coo:       # X as lists, R dense: the lists of its non-zeros
            Rd = SparseTensor.fromarray(np.asarray(Rd) != 0)
        elif Rd is not None and not _is_torch(Rd):
            Rd = to_dense_u8(Rd, "R")
            if Rd.dtype != np.uint8 or Rd.max(initial=0) > 1:
                Rd = (Rd != 0).astype(np.uint8)

        self.EPS = float(extra["EPS"]) if "EPS" in extra else DEFAULT_EPS
        self.bias0 = float(extra["bias0"]) if "bias0" in extra else DEFAULT_BIAS0
        self.max_iter = int(extra["max_iter"]) if "max_iter" in extra else DEFAULT_MAX_ITER
        self.num_realisations = int(extra["num_realisations"]) if "num_realisations" in extra else DEFAULT_NUM_REALISATIONS

        if "alpha_theta" in extra or "beta_theta" in extra:
            self.alpha_theta, self.beta_theta = extra["alpha_theta"], extra["beta_theta"]
            if np.shape(self.alpha_theta) != (self.L, self.M):
                msg = "alpha_theta matrix is not valid. When using this parameter, make sure to inform a %d x %d matrix."
                self.logger.error(msg)
                raise ValueError(msg % (self.L, self.M))
            if np.shape(self.beta_theta) != (self.L, self.M):
                msg = "beta_theta matrix is not valid. When using this parameter, make sure to inform a %d x %d matrix."
                self.logger.error(msg)
                raise ValueError(msg % (self.L, self.M))
        else:
            if type(theta_prior) is not tuple or len(theta_prior) != 2:
                msg = "theta_prior must be a 2D tuple!"
                self.logger.error(msg)
                raise ValueError(msg)
            self.alpha_theta, self.beta_theta = theta_prior

        if "alpha_lambda" in extra or "beta_lambda" in extra:
            self.alpha_lambda, self.beta_lambda = extra["alpha_lambda"], extra["beta_lambda"]
            for nm, arr in (("alpha_lambda", self.alpha_lambda), ("beta_lambda", self.beta_lambda)):
                if np.shape(arr) != (self.L, self.K):
                    sh = np.shape(arr)
                    msg = f"{nm} matrix is not valid (dimensions = %d x %d)."
                    msg += "When using this parameter, make sure to pass a %d x %d matrix."
                    msg = msg % (sh[0] if len(sh) > 0 else 0, sh[1] if len(sh) > 1 else 0, self.L, self.K)
                    self.logger.error(msg)
                    raise ValueError(msg)
        else:
            if type(lambda_prior) is not tuple or len(lambda_prior) != 2:
                msg = "lambda_prior must be a 2D tuple!"
                self.logger.error(msg)
                raise ValueError(msg)
            self.alpha_lambda, self.beta_lambda = lambda_prior

        if type(eta_prior) is not tuple or len(eta_prior) != 2:
            msg = "eta_prior must be a 2D tuple!"
            self.logger.error(msg)
            raise ValueError(msg)
        self.alpha_mutuality, self.beta_mutuality = eta_prior

        if rho_prior is not None and np.shape(rho_prior) != (self.L, self.N, self.N):
            msg = "rho_prior has to have shape equal to (L, N, N)!"
            self.logger.error(msg)
            raise ValueError(msg)
        self.rho_prior = rho_prior
        self._change_seed(seed)
        return Xd, Rd

    def _change_seed(self, seed):
        self.seed = seed
        self.prng = np.random.RandomState(seed)

    # ------------------------------------------------------------------ initial state (model.py:458-605)
    def _draw_pr_rho(self, coverage, bias0, prng=None, out=None):
        """`_set_rho_prior` (model.py:458-559).  The common case (no informative prior, directed network) runs as one C
        pass that is bit-identical to the NumPy statements below (vimure_amd/csrc/host_init.c), into `out` if given."""
        prng = self.prng if prng is None else prng
        L, N, K = self.L, self.N, self.K
        if self.rho_prior is None and not self.undirected:
            from . import _hostlib
            pr = _hostlib.draw_pr_rho(prng, (L, N, N, K), bias0, coverage, out=out)
            if pr is not None:
                return pr
        if self.rho_prior is None:
            pr = 1.0 + 0.01 * prng.rand(L, N, N, K)
            pr[..., 0] += bias0
            if self.undirected:
                pr = (pr + pr.transpose(0, 2, 1, 3)) / 2.0
            pr /= pr.sum(axis=-1)[..., None]
        else:
            pr = np.zeros((L, N, N, K))
            sub = np.nonzero(self.rho_prior)
            n = sub[0].shape[0]
            for k in range(K):
                pr[sub + (np.full(n, k),)] = poisson.pmf(k, self.rho_prior[sub]) + 1.0 * prng.rand(n)
            if self.undirected:
                pr = (pr + pr.transpose(0, 2, 1, 3)) / 2.0
            pr[sub] /= pr[sub].sum(axis=-1)[:, None]
        onehot = np.zeros(K)
        onehot[0] = 1.0
        pr[coverage == 0] = onehot  # ties no reporter covers / nobody reported (model.py:508-556)
        if out is not None:
            out[...] = pr
            return out
        return pr

    def _draw_gammas(self, sumX, prng=None):
        """`_initialize_priors` (model.py:561-605): the draws that follow the rho prior, in the reference's order."""
        prng = self.prng if prng is None else prng
        L, M, K = self.L, self.M, self.K
        st = {}
        st["gamma_shp"] = self.alpha_theta * prng.random_sample(size=(L, M)) + self.alpha_theta
        st["phi_shp"] = self.alpha_lambda * prng.random_sample(size=(L, K)) + self.alpha_lambda
        st["gamma_rte"] = self.beta_theta * prng.random_sample(size=(L, M)) + self.beta_theta
        st["phi_rte"] = self.beta_lambda * prng.random_sample(size=(L, K)) + self.beta_lambda
        if self.mutuality:
            st["nu_shp"] = self.alpha_mutuality * prng.random_sample(1)[0] + self.alpha_mutuality
            st["nu_rte"] = self.beta_mutuality + sumX  # fixed once and for all (model.py:593-595)
        else:
            st["nu_shp"], st["nu_rte"] = 0.000001, 1.0
        if prng is self.prng:
            for k, v in st.items():
                setattr(self, k, v)
        return st

    def _staging_index(self, eng, r):
        """Which pinned staging buffer realisation r is drawn into: with upload-ahead (see fit) buffer 1 is free again as soon as
        its copy to the device is done, so two buffers serve any number of realisations; otherwise three in rotation."""
        if self.num_realisations > 1 and getattr(eng, "can_upload_ahead", lambda: False)():
            return 0 if r == 0 else 1
        return r % 3

    def _initial_states(self, eng, coverage):
        """Initial state of every realisation, in order: (r, seed of r, state dict incl. pr_rho, seed after r).  CAVI
        consumes no randomness (reference model.py:386-437), so the whole seed chain is a function of the first seed
        and realisation r + 1 can be drawn while r runs on the GPU.  pr_rho lands in the engine's staging buffers
        (three in rotation: one being uploaded, one queued, one being drawn)."""
        seed, prng = self.seed, self.prng
        for r in range(self.num_realisations):
            bias = DEFAULT_BIAS0 if r < 5 else (r - 4) * self.bias0
            pr = self._draw_pr_rho(coverage, bias, prng=prng, out=eng.staging(self._staging_index(eng, r)))
            st = self._draw_gammas(self.sumX, prng=prng)
            st["pr_rho"] = pr
            step = prng.randint(1, 500)
            nxt = step if seed is None else seed + step
            yield r, seed, st, nxt
            seed, prng = nxt, np.random.RandomState(nxt)

    # ------------------------------------------------------------------ fit (model.py:327-448)
    def fit(self, X, theta_prior=(0.1, 0.1), lambda_prior=(10.0, 10.0), eta_prior=(0.5, 1.0), rho_prior=None,
            seed: int = None, **extra_params):
        """Same contract as the reference's `fit`; extra keywords: `device` picks the GPU (default 0, or the
        device of a torch tensor X); `engine` reuses a `CaviEngine` already holding this X, R, K (many seeds
        of one dataset: the data is uploaded once, see vimure_amd/batch.py); `keep_engine=True` leaves the posteriors on
        the GPU after the fit: `get_inferred_model` / `predict` then run there (vmr_readout) and `rho_f` is only copied
        to the host if something asks for it (`close()` frees the device memory).

        Host work per realisation is the RandomState draw of the initial state (bit-exact with the reference); with
        several realisations the next draw runs on a host thread while the GPU sweeps, the best realisation is kept on
        the device (vmr_snapshot) and rho crosses PCIe once, at the end.  `rho` (without `_f`) holds the best
        realisation too (the reference leaves the last one there; nothing reads it after `fit`)."""
        Xd, Rd = self._check_fit_params(X, lambda_prior, theta_prior, eta_prior, rho_prior, seed, **extra_params)
        self.close()   # a device state kept by an earlier fit(keep_engine=True)
        self._engine, self._rho_f = None, None
        eng = extra_params.get("engine")
        own_engine = eng is None
        keep = bool(extra_params.get("keep_engine", False)) and own_engine
        if own_engine and is_sparse_like(Xd):
            eng = CaviEngine.from_coo(Xd.subs, Xd.vals, (self.L, self.N, self.N, self.M), R=None if Rd is None else Rd.subs,
                                      K=self.K, mutuality=self.mutuality, eps=self.EPS, device=extra_params.get("device"))
        elif own_engine:
            eng = CaviEngine(Xd, Rd, K=self.K, mutuality=self.mutuality, eps=self.EPS, device=extra_params.get("device"))
        elif (eng.L, eng.N, eng.M, eng.K, eng.mutuality) != (self.L, self.N, self.M, self.K, bool(self.mutuality)):
            raise ValueError("engine does not match the shape / K / mutuality of this fit")
        producer = None
        try:
            self.sumX, coverage = eng.data_stats()
            eng.set_priors(self.alpha_theta, self.beta_theta, self.alpha_lambda, self.beta_lambda,
                           self.alpha_mutuality, self.beta_mutuality)
            maxL, trace, best = -INF, [], None
            self.loop_seconds = 0.0   # wall time inside the CAVI loops of all realisations (device work included)
            self.draw_seconds = 0.0   # host time the loops waited for an initial state
            states = self._initial_states(eng, coverage)
            if self.num_realisations > 1:   # draw realisation r + 1 while r runs
                import queue
                import threading
                q = queue.Queue(maxsize=1)   # one finished state waits while the next is drawn (three staging buffers)
                stop = threading.Event()     # set on any exit from fit(): the producer draws nothing further
                # upload-ahead keeps two device slots in rotation: slot r % 2 may only be overwritten once set_state(r) has
                # consumed it (set_state synchronises the engine's stream before it returns)
                slot_free = [threading.Event(), threading.Event()]
                for ev in slot_free:
                    ev.set()

                def put(item):
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            return True
                        except queue.Full:
                            continue
                    return False

                def work():
                    try:
                        for item in states:
                            if stop.is_set():
                                return
                            r_ = item[0]
                            pr_ = item[2]["pr_rho"]
                            si = self._staging_index(eng, r_)
                            if r_ > 0 and getattr(eng, "can_upload_ahead", lambda: False)() and isinstance(pr_, np.ndarray) and np.shares_memory(pr_, eng.staging(si)):
                                slot = r_ % 2
                                while not slot_free[slot].wait(timeout=0.1):
                                    if stop.is_set():
                                        return
                                slot_free[slot].clear()
                                dev = eng.upload_ahead(si, slot)   # (realisation 0 is waited for: nothing to hide its upload behind)
                                if dev is not None:
                                    item[2]["pr_rho"] = dev
                                    item[2]["_slot"] = slot
                                else:
                                    slot_free[slot].set()
                            if not put(item):
                                return
                        put(None)
                    except BaseException as e:   # surfaces in the consumer
                        put(e)
                producer = threading.Thread(target=work, daemon=True)
                producer.start()

                def next_state():
                    item = q.get()
                    if isinstance(item, BaseException):
                        raise item
                    return item
            else:
                def next_state():
                    return next(states, None)
            final_seed = self.seed
            while True:
                t_draw = time.perf_counter()
                item = next_state()
                self.draw_seconds += time.perf_counter() - t_draw
                if item is None:
                    break
                r, seed_r, st, final_seed = item
                eng.set_state(st["gamma_shp"], st["gamma_rte"], st["phi_shp"], st["phi_rte"], st["nu_shp"], st["nu_rte"],
                              st["pr_rho"])   # (synchronises: the staging buffer is free again)
                if "_slot" in st:   # the device slot of upload_ahead has been read: the producer may reuse it
                    slot_free[st["_slot"]].set()
                t_loop = time.perf_counter()
                if not self.verbose:   # the whole loop on the engine's side (vmr_fit_loop): one call per realisation
                    rows, elbo, _, _ = eng.fit_loop(self.max_iter, self.convergence_tol, self.decision)
                    trace.extend((r, seed_r, it_, e_, rt_, rc_) for it_, e_, rt_, rc_ in rows)
                else:
                    elbo = self._loop_verbose(eng, r, seed_r, trace)
                eng.sync()
                self.loop_seconds += time.perf_counter() - t_loop
                self._pull_params(eng)   # the small arrays of this realisation (rho stays on the device)
                if maxL < elbo:
                    maxL, best = elbo, self._params_copy()
                    if self.num_realisations > 1:
                        eng.snapshot()
            if best is not None:
                if self.num_realisations > 1:
                    eng.restore()
                self._update_optimal_parameters(best, eng, lazy_rho=keep)
            self._change_seed(final_seed)
            if keep:
                self._engine, own_engine = eng, False
        finally:
            if producer is not None:   # stop the producer and wait for it BEFORE the engine (and its staging buffers) go away
                stop.set()
                while producer.is_alive():
                    try:
                        q.get(timeout=0.05)
                    except queue.Empty:
                        pass
                producer.join()
            if own_engine:
                eng.close()
        cols = ["realisation", "seed", "iter", "elbo", "runtime", "reached_convergence"]
        self.trace = pd.DataFrame(trace, columns=cols)
        self.maxL = maxL
        return self

    def _loop_verbose(self, eng, r, seed_r, trace):
        """The reference's while-loop (model.py:405-426) step by step, with its DEBUG line per ELBO evaluation."""
        coincide, it, reached, elbo = 0, 1, False, -INF
        while not reached and it <= self.max_iter:
            nxt = it if (it == 1 or it % 10 == 0 or it == self.max_iter) else min(self.max_iter, (it // 10 + 1) * 10)
            if nxt > it:
                eng.step(nxt - it)
                it = nxt
                eng.sync()
            t0 = time.time()
            old, elbo = elbo, eng.step(1, want_elbo=True)
            runtime = time.time() - t0
            coincide = coincide + 1 if abs(elbo - old) < self.convergence_tol else 0
            if coincide > self.decision:
                reached = True
            self.logger.debug(f"Realisation {r:2} | Iter {it:4} | ELBO value: {elbo:6.12f} | Reached convergence: {reached}")
            it += 1
            if (it - 1) % 10 == 0:
                trace.append((r, seed_r, it - 1, elbo, runtime, reached))
        return elbo

    _SMALL = ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte", "nu_shp", "nu_rte")

    def _pull_params(self, eng):
        st = eng.get_state(rho=False)
        self.gamma_shp, self.gamma_rte = st["gamma_shp"], st["gamma_rte"]
        self.phi_shp, self.phi_rte = st["phi_shp"], st["phi_rte"]
        self.nu_shp, self.nu_rte = np.float64(st["nu_shp"]), np.float64(st["nu_rte"])
        self.G_exp_theta = np.exp(sp.psi(self.gamma_shp) - np.log(self.gamma_rte))
        self.G_exp_lambda = np.exp(sp.psi(self.phi_shp) - np.log(self.phi_rte))
        # what the last cache refresh held, i.e. computed before the last nu update (model.py:684 vs :822)
        self.G_exp_nu = np.float64(eng.get_geometric()[3]) if self.mutuality else 0.0

    def _params_copy(self):
        return {k: np.copy(getattr(self, k)) for k in self._SMALL}

    def _update_optimal_parameters(self, best, eng, lazy_rho=False):
        """model.py:925-942; rho of the best realisation is the engine's current state here (vmr_restore)."""
        self.gamma_shp_f, self.gamma_rte_f = best["gamma_shp"], best["gamma_rte"]
        self.phi_shp_f, self.phi_rte_f = best["phi_shp"], best["phi_rte"]
        self.nu_shp_f, self.nu_rte_f = best["nu_shp"], best["nu_rte"]
        self._rho_f = None
        if not lazy_rho:
            self._fetch_rho(eng)
        self.G_exp_theta_f = np.exp(sp.psi(self.gamma_shp_f) - np.log(self.gamma_rte_f))
        self.G_exp_lambda_f = np.exp(sp.psi(self.phi_shp_f) - np.log(self.phi_rte_f))
        self.G_exp_nu_f = np.exp(sp.psi(self.nu_shp_f) - np.log(self.nu_rte_f))

    # rho_f / rho: on the host once fetched; with fit(keep_engine=True) still on the GPU until something reads them
    def _fetch_rho(self, eng):
        buf, keepalive = host_buffer(self.L * self.N * self.N * self.K)
        self._rho_f = eng.get_state(rho=True, rho_out=buf)["rho"]
        self._rho_keepalive = keepalive   # page-locked memory behind rho_f, when it is
        return self._rho_f

    @property
    def rho_f(self):
        if getattr(self, "_rho_f", None) is None:
            eng = getattr(self, "_engine", None)
            if eng is None:
                raise AttributeError("rho_f: the model has not been fitted")
            self._fetch_rho(eng)
        return self._rho_f

    @rho_f.setter
    def rho_f(self, value):
        self._rho_f = value

    @property
    def rho(self):
        return self.rho_f

    def close(self):
        """Free the device state kept by fit(keep_engine=True); rho_f is fetched first if nothing has read it yet."""
        eng = getattr(self, "_engine", None)
        if eng is not None:
            if getattr(self, "_rho_f", None) is None:
                self._fetch_rho(eng)
            eng.close()
            self._engine = None

    def __del__(self):
        eng = getattr(self, "_engine", None)
        if eng is not None:
            try:
                eng.close()
            except Exception:
                pass

    # ------------------------------------------------------------------ read-out (model.py:1062-1214)
    def sample_inferred_model(self, N=1, seed=None, device=False):
        """Reference model.py:1062-1096: N samples of Y, sample i = `default_rng(seed + i).multinomial(N, rho_f).argmax(-1)`.
        device=True (after `fit(keep_engine=True)`): the same draw on the GPU from the rho kept there (vmr_sample) -- per tie
        the most frequent category of N categorical trials, a Philox stream keyed by seed + i instead of NumPy's PCG64 (same
        distribution, reproducible for a seed, different numbers) -- so rho (8 L N^2 K bytes) never crosses PCIe."""
        if seed is None:
            seed = self.seed
        if device:
            eng = getattr(self, "_engine", None)
            if eng is None:
                raise ValueError("device=True needs the posteriors on the GPU: fit(..., keep_engine=True)")
            return [eng.sample(seed + i, n_trials=N).astype(np.int64) for i in range(N)]

        def sample_y(s):
            g = np.random.default_rng(s)
            return g.multinomial(n=N, pvals=self.rho_f, size=(self.L, self.N, self.N)).argmax(axis=-1)

        return [sample_y(seed + i) for i in range(N)]

    def get_inferred_model(self, method="rho_max", threshold=None):
        options = ["rho_max", "rho_mean", "fixed_threshold", "heuristic_threshold"]
        if method not in options:
            raise ValueError("'method' should be one of {}.".format(", ".join(['"' + x + '"' for x in options])))
        if (not self.mutuality and method != "rho_max") or (self.K > 2 and "threshold" in method):
            msg = ('threshold methods is incompatible with VIMuRe\'s mutuality=False '
                   'or for data with more than 2 categories. Using "rho_max" method.')
            warnings.warn(msg, UserWarning)
            method = "rho_max"
        # rho still on the GPU (fit(keep_engine=True), nothing has read rho_f): the read-out runs there (vmr_readout)
        dev = getattr(self, "_engine", None) if getattr(self, "_rho_f", None) is None else None
        if method == "rho_max":
            if dev is not None:
                return dev.readout("rho_max").astype("int")
            return np.argmax(self.rho_f, axis=-1).astype("int")
        if method == "rho_mean":
            if dev is not None:
                return dev.readout("rho_mean")
            return np.dot(self.rho_f, range(0, self.rho_f.shape[-1]))
        if method == "fixed_threshold":
            if threshold is None or threshold > 1 or threshold < 0:
                raise ValueError('For method="fixed_threshold", you must set the threshold to a value in [0,1].')
        else:  # heuristic threshold, reference utils.py:200-217
            threshold = 0.54 * self.G_exp_nu - 0.01
        if dev is not None:
            Y = dev.readout("threshold", threshold)
            return Y.astype(np.float64) if method == "fixed_threshold" else Y.astype("int")
        Y = np.copy(self.rho_f[:, :, :, 1])
        Y[Y < threshold] = 0
        Y[Y >= threshold] = 1
        return Y if method == "fixed_threshold" else Y.astype("int")

    def predict(self, X=None, method="rho_max", threshold=None):
        """Alias of `get_inferred_model` (the reference's experiment wrapper calls it predict)."""
        return self.get_inferred_model(method=method, threshold=threshold)

    def get_posterior_estimates(self):
        return {"nu": self.G_exp_nu_f, "theta": self.G_exp_theta_f, "lambda": self.G_exp_lambda_f, "rho": self.rho_f}
