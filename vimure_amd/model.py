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
from .engine import CaviEngine
from .tensor import is_sparse_like, to_dense_u8

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
_OURS = ["device", "alpha_theta", "beta_theta", "engine"]


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
        if dev_tensor:
            Xd = X
            shape = tuple(int(s) for s in X.shape)
        else:
            Xd = to_dense_u8(X, "X")
            shape = Xd.shape
        if len(shape) != 4 or shape[1] != shape[2]:
            raise ValueError("X must have shape (L, N, N, M)")
        self.L, self.N, self.M = shape[0], shape[1], shape[3]

        if self.undirected:
            sym = bool((Xd == Xd.transpose(1, 2)).all()) if dev_tensor else np.array_equal(Xd, Xd.transpose(0, 2, 1, 3))
            if not sym:
                msg = "If undirected is True, the given network has to be symmetric wrt l and m!"
                self.logger.error(msg)
                raise ValueError(msg)

        if not hasattr(self, "K"):
            if extra.get("K") is not None:
                self.K = int(extra["K"])
            else:
                self.K = int(Xd.max()) + 1
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
        if Rd is not None and not _is_torch(Rd):
            Rd = to_dense_u8(Rd, "R")
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
    def _draw_pr_rho(self, coverage, bias0):
        L, N, K = self.L, self.N, self.K
        if self.rho_prior is None:
            pr = 1.0 + 0.01 * self.prng.rand(L, N, N, K)
            pr[..., 0] += bias0
            if self.undirected:
                pr = (pr + pr.transpose(0, 2, 1, 3)) / 2.0
            pr /= pr.sum(axis=-1)[..., None]
        else:
            pr = np.zeros((L, N, N, K))
            sub = np.nonzero(self.rho_prior)
            n = sub[0].shape[0]
            for k in range(K):
                pr[sub + (np.full(n, k),)] = poisson.pmf(k, self.rho_prior[sub]) + 1.0 * self.prng.rand(n)
            if self.undirected:
                pr = (pr + pr.transpose(0, 2, 1, 3)) / 2.0
            pr[sub] /= pr[sub].sum(axis=-1)[:, None]
        onehot = np.zeros(K)
        onehot[0] = 1.0
        pr[coverage == 0] = onehot  # ties no reporter covers / nobody reported (model.py:508-556)
        return pr

    def _draw_gammas(self, sumX):
        L, M, K = self.L, self.M, self.K
        self.gamma_shp = self.alpha_theta * self.prng.random_sample(size=(L, M)) + self.alpha_theta
        self.phi_shp = self.alpha_lambda * self.prng.random_sample(size=(L, K)) + self.alpha_lambda
        self.gamma_rte = self.beta_theta * self.prng.random_sample(size=(L, M)) + self.beta_theta
        self.phi_rte = self.beta_lambda * self.prng.random_sample(size=(L, K)) + self.beta_lambda
        if self.mutuality:
            self.nu_shp = self.alpha_mutuality * self.prng.random_sample(1)[0] + self.alpha_mutuality
            self.nu_rte = self.beta_mutuality + sumX  # fixed once and for all (model.py:593-595)
        else:
            self.nu_shp, self.nu_rte = 0.000001, 1.0

    # ------------------------------------------------------------------ fit (model.py:327-448)
    def fit(self, X, theta_prior=(0.1, 0.1), lambda_prior=(10.0, 10.0), eta_prior=(0.5, 1.0), rho_prior=None,
            seed: int = None, **extra_params):
        """Same contract as the reference's `fit`; extra keywords: `device` picks the GPU (default 0, or the
        device of a torch tensor X); `engine` reuses a `CaviEngine` already holding this X, R, K (many seeds
        of one dataset: the data is uploaded once, see vimure_amd/batch.py)."""
        Xd, Rd = self._check_fit_params(X, lambda_prior, theta_prior, eta_prior, rho_prior, seed, **extra_params)
        eng = extra_params.get("engine")
        own_engine = eng is None
        if own_engine:
            eng = CaviEngine(Xd, Rd, K=self.K, mutuality=self.mutuality, eps=self.EPS, device=extra_params.get("device"))
        elif (eng.L, eng.N, eng.M, eng.K, eng.mutuality) != (self.L, self.N, self.M, self.K, bool(self.mutuality)):
            raise ValueError("engine does not match the shape / K / mutuality of this fit")
        try:
            self.sumX, coverage = eng.data_stats()
            eng.set_priors(self.alpha_theta, self.beta_theta, self.alpha_lambda, self.beta_lambda,
                           self.alpha_mutuality, self.beta_mutuality)
            maxL, trace = -INF, []
            self.loop_seconds = 0.0   # wall time inside the CAVI loops of all realisations (device work included)
            for r in range(self.num_realisations):
                bias = DEFAULT_BIAS0 if r < 5 else (r - 4) * self.bias0
                pr_rho = self._draw_pr_rho(coverage, bias)
                self._draw_gammas(self.sumX)
                eng.set_state(self.gamma_shp, self.gamma_rte, self.phi_shp, self.phi_rte, self.nu_shp, self.nu_rte,
                              pr_rho)
                del pr_rho
                coincide, it, reached, elbo = 0, 1, False, -INF
                t_loop = time.perf_counter()
                while not reached and it <= self.max_iter:
                    check = it == 1 or it % 10 == 0 or it == self.max_iter
                    t0 = time.time()
                    if check:
                        old, elbo = elbo, eng.step(1, want_elbo=True)
                        coincide = coincide + 1 if abs(elbo - old) < self.convergence_tol else 0
                    else:
                        eng.step(1)
                    runtime = time.time() - t0
                    if coincide > self.decision:
                        reached = True
                    if check and self.verbose:
                        self.logger.debug(f"Realisation {r:2} | Iter {it:4} | ELBO value: {elbo:6.12f} | "
                                          f"Reached convergence: {reached}")
                    it += 1
                    if (it - 1) % 10 == 0:
                        trace.append((r, self.seed, it - 1, elbo, runtime, reached))
                eng.sync()
                self.loop_seconds += time.perf_counter() - t_loop
                self._pull_state(eng)
                if maxL < elbo:
                    self._update_optimal_parameters()
                    maxL = elbo
                step = self.prng.randint(1, 500)
                self._change_seed(step if self.seed is None else self.seed + step)
        finally:
            if own_engine:
                eng.close()
        cols = ["realisation", "seed", "iter", "elbo", "runtime", "reached_convergence"]
        self.trace = pd.DataFrame(trace, columns=cols)
        self.maxL = maxL
        return self

    def _pull_state(self, eng):
        st = eng.get_state(rho=True)
        self.gamma_shp, self.gamma_rte = st["gamma_shp"], st["gamma_rte"]
        self.phi_shp, self.phi_rte = st["phi_shp"], st["phi_rte"]
        self.nu_shp, self.nu_rte = np.float64(st["nu_shp"]), np.float64(st["nu_rte"])
        self.rho = st["rho"]
        self.G_exp_theta = np.exp(sp.psi(self.gamma_shp) - np.log(self.gamma_rte))
        self.G_exp_lambda = np.exp(sp.psi(self.phi_shp) - np.log(self.phi_rte))
        # what the last cache refresh held, i.e. computed before the last nu update (model.py:684 vs :822)
        self.G_exp_nu = np.float64(eng.get_geometric()[3]) if self.mutuality else 0.0

    def _update_optimal_parameters(self):
        """model.py:925-942"""
        self.gamma_shp_f, self.gamma_rte_f = np.copy(self.gamma_shp), np.copy(self.gamma_rte)
        self.phi_shp_f, self.phi_rte_f = np.copy(self.phi_shp), np.copy(self.phi_rte)
        self.nu_shp_f, self.nu_rte_f = np.copy(self.nu_shp), np.copy(self.nu_rte)
        self.rho_f = np.copy(self.rho)
        self.G_exp_theta_f = np.exp(sp.psi(self.gamma_shp_f) - np.log(self.gamma_rte_f))
        self.G_exp_lambda_f = np.exp(sp.psi(self.phi_shp_f) - np.log(self.phi_rte_f))
        self.G_exp_nu_f = np.exp(sp.psi(self.nu_shp_f) - np.log(self.nu_rte_f))

    # ------------------------------------------------------------------ read-out (model.py:1062-1214)
    def sample_inferred_model(self, N=1, seed=None):
        if seed is None:
            seed = self.seed

        def sample_y(s):
            g = np.random.default_rng(s)
            return g.multinomial(n=N, pvals=self.rho_f, size=(self.L, self.N, self.N)).argmax(axis=-1)

        return [sample_y(seed + i) for i in range(N)]

    def get_inferred_model(self, method="rho_max", threshold=None):
        options = ["rho_max", "rho_mean", "fixed_threshold", "heuristic_threshold"]
        if method not in options:
            raise ValueError("'method' should be one of {}.".format(", ".join(['"' + x + '"' for x in options])))
        if (not self.mutuality and method != "rho_max") or (self.rho_f.shape[-1] > 2 and "threshold" in method):
            msg = ('threshold methods is incompatible with VIMuRe\'s mutuality=False '
                   'or for data with more than 2 categories. Using "rho_max" method.')
            warnings.warn(msg, UserWarning)
            method = "rho_max"
        if method == "rho_max":
            return np.argmax(self.rho_f, axis=-1).astype("int")
        if method == "rho_mean":
            return np.dot(self.rho_f, range(0, self.rho_f.shape[-1]))
        if method == "fixed_threshold":
            if threshold is None or threshold > 1 or threshold < 0:
                raise ValueError('For method="fixed_threshold", you must set the threshold to a value in [0,1].')
        else:  # heuristic threshold, reference utils.py:200-217
            threshold = 0.54 * self.G_exp_nu - 0.01
        Y = np.copy(self.rho_f[:, :, :, 1])
        Y[Y < threshold] = 0
        Y[Y >= threshold] = 1
        return Y if method == "fixed_threshold" else Y.astype("int")

    def predict(self, X=None, method="rho_max", threshold=None):
        """Alias of `get_inferred_model` (the reference's experiment wrapper calls it predict)."""
        return self.get_inferred_model(method=method, threshold=threshold)

    def get_posterior_estimates(self):
        return {"nu": self.G_exp_nu_f, "theta": self.G_exp_theta_f, "lambda": self.G_exp_lambda_f, "rho": self.rho_f}
