"""CPU oracle for the VIMuRe CAVI hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A dense-tensor NumPy restatement of the reference algorithm
(`/root/reference/src/python/vimure/model.py`), written from the equations in
SURVEY.md section 3.2-3.4, not from the reference's loops.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this module; the product package `vimure_amd` never does.

Parity pin: `tools/make_golden.py` imports the real reference (in the build
container only) and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks every function below against those vectors.

Data layout: X  uint8/int [L,N,N,M]  (layer, ego, alter, reporter) counts,
             R  uint8/bool [L,N,N,M] reporter mask (0/1),
             rho float64 [L,N,N,K].
Everything is float64, as in the reference.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
from scipy.special import gammaln, psi
from scipy.stats import poisson

INF = 1e10
DEFAULT_EPS = 1e-12


@dataclass
class Priors:
    """Gamma hyper-parameters, broadcast to full arrays (model.py:238-317)."""
    alpha_theta: np.ndarray  # [L,M]
    beta_theta: np.ndarray   # [L,M]
    alpha_lambda: np.ndarray  # [L,K]
    beta_lambda: np.ndarray   # [L,K]
    alpha_eta: float
    beta_eta: float


def make_priors(L, M, K, theta_prior=(0.1, 0.1), lambda_prior=(10.0, 10.0), eta_prior=(0.5, 1.0),
                alpha_theta=None, beta_theta=None, alpha_lambda=None, beta_lambda=None) -> Priors:
    at = np.broadcast_to(np.asarray(theta_prior[0] if alpha_theta is None else alpha_theta, float), (L, M)).copy()
    bt = np.broadcast_to(np.asarray(theta_prior[1] if beta_theta is None else beta_theta, float), (L, M)).copy()
    al = np.broadcast_to(np.asarray(lambda_prior[0] if alpha_lambda is None else alpha_lambda, float), (L, K)).copy()
    bl = np.broadcast_to(np.asarray(lambda_prior[1] if beta_lambda is None else beta_lambda, float), (L, K)).copy()
    return Priors(at, bt, al, bl, float(eta_prior[0]), float(eta_prior[1]))


@dataclass
class State:
    """Variational state (model.py:561-617) for one realisation."""
    gamma_shp: np.ndarray
    gamma_rte: np.ndarray
    phi_shp: np.ndarray
    phi_rte: np.ndarray
    nu_shp: float
    nu_rte: float
    rho: np.ndarray
    pr_rho: np.ndarray
    logpr_rho: np.ndarray
    # exp(E[log nu]) as last computed by a cache refresh (model.py:684 / :596-600);
    # the ELBO uses this stale value.
    G_nu_cache: float = 0.0


@dataclass
class Problem:
    """Data + static configuration for a fit."""
    X: np.ndarray
    R: np.ndarray
    K: int
    mutuality: bool
    priors: Priors
    eps: float = DEFAULT_EPS
    undirected: bool = False
    # derived
    nz: tuple = field(default=None, repr=False)
    x_nz: np.ndarray = field(default=None, repr=False)
    xT_nz: np.ndarray = field(default=None, repr=False)
    sumX: float = 0.0

    def __post_init__(self):
        self.X = np.asarray(self.X)
        self.R = np.asarray(self.R)
        assert self.X.ndim == 4 and self.X.shape == self.R.shape
        self.L, self.N, _, self.M = self.X.shape
        self.nz = np.nonzero(self.X)
        self.x_nz = self.X[self.nz].astype(np.float64)
        l, i, j, m = self.nz
        self.xT_nz = self.X[l, j, i, m].astype(np.float64)
        self.sumX = float(self.X.sum(dtype=np.int64))


# --------------------------------------------------------------------------
# initialisation (model.py:450-617)
# --------------------------------------------------------------------------

def init_pr_rho(pb: Problem, prng: np.random.RandomState, bias0: float = 0.0,
                rho_prior: Optional[np.ndarray] = None) -> np.ndarray:
    """Prior on rho (model.py:458-559): 1 + 0.01*U, normalised; one-hot(k=0) for
    ties nobody may report on (no R entry) or nobody reported (no X entry)."""
    L, N, K = pb.L, pb.N, pb.K
    if rho_prior is None:
        pr = 1.0 + 0.01 * prng.rand(L, N, N, K)
        pr[..., 0] += bias0
        if pb.undirected:
            pr = (pr + pr.transpose(0, 2, 1, 3)) / 2.0
        pr /= pr.sum(axis=-1)[..., None]
    else:
        pr = np.zeros((L, N, N, K))
        sub = np.nonzero(rho_prior)
        n = sub[0].shape[0]
        for k in range(K):
            pr[sub + (np.full(n, k),)] = poisson.pmf(k, rho_prior[sub]) + prng.rand(n)
        if pb.undirected:
            pr = (pr + pr.transpose(0, 2, 1, 3)) / 2.0
        pr[sub] /= pr[sub].sum(axis=-1)[:, None]
    covered = pb.R.any(axis=3) & (pb.X != 0).any(axis=3)
    onehot = np.zeros(K)
    onehot[0] = 1.0
    pr[~covered] = onehot
    return pr


def init_state(pb: Problem, prng: np.random.RandomState, bias0: float = 0.0,
               rho_prior: Optional[np.ndarray] = None) -> State:
    """Random start (model.py:561-605); draw order fixes the seed chain."""
    L, M, K, p = pb.L, pb.M, pb.K, pb.priors
    pr = init_pr_rho(pb, prng, bias0, rho_prior)
    gs = p.alpha_theta * prng.random_sample(size=(L, M)) + p.alpha_theta
    ps = p.alpha_lambda * prng.random_sample(size=(L, K)) + p.alpha_lambda
    gr = p.beta_theta * prng.random_sample(size=(L, M)) + p.beta_theta
    pr_ = p.beta_lambda * prng.random_sample(size=(L, K)) + p.beta_lambda
    if pb.mutuality:
        ns = p.alpha_eta * prng.random_sample(1)[0] + p.alpha_eta
        nr = p.beta_eta + pb.sumX
        g_nu = float(np.exp(psi(ns) - np.log(nr)))
    else:
        ns, nr, g_nu = 1e-6, 1.0, 0.0
    return State(gs, gr, ps, pr_, float(ns), float(nr), pr.copy(), pr, np.log(pr + pb.eps), g_nu)


def next_seed(seed, prng: np.random.RandomState):
    """Seed of the following realisation (model.py:432-437)."""
    step = prng.randint(1, 500)
    return step if seed is None else seed + step


# --------------------------------------------------------------------------
# expectations
# --------------------------------------------------------------------------

def _mean(shp, rte):
    return shp / rte


def _elog(shp, rte):
    return psi(shp) - np.log(rte)


def _weights(pb: Problem, st: State):
    """Per-nonzero weights of a cache refresh (model.py:662-696): returns
    (x*w1 [I,K or I,1], x*w2 [I,K] or None) and stores the refreshed G_nu."""
    if not pb.mutuality:
        return pb.x_nz[:, None], None
    l, _, _, m = pb.nz
    G_th = np.exp(_elog(st.gamma_shp, st.gamma_rte))
    G_la = np.exp(_elog(st.phi_shp, st.phi_rte))
    st.G_nu_cache = float(np.exp(psi(st.nu_shp) - np.log(st.nu_rte)))
    z1 = G_th[l, m][:, None] * G_la[l, :]
    z2 = (st.G_nu_cache * pb.xT_nz)[:, None]
    den = z1 + z2
    den[den == 0] = 1.0
    return pb.x_nz[:, None] * z1 / den, pb.x_nz[:, None] * z2 / den


# --------------------------------------------------------------------------
# the four coordinate updates (model.py:698-830)
# --------------------------------------------------------------------------

def update_gamma(pb: Problem, st: State) -> None:
    l, i, j, m = pb.nz
    xw1, _ = _weights(pb, st)
    contrib = (st.rho[l, i, j, :] * xw1).sum(axis=1)
    shp = np.zeros((pb.L, pb.M))
    np.add.at(shp, (l, m), contrib)
    e = np.einsum("lijk,lk->lij", st.rho, _mean(st.phi_shp, st.phi_rte))
    rte = np.einsum("lij,lijm->lm", e, pb.R.astype(np.float64))
    st.gamma_shp = pb.priors.alpha_theta + shp
    st.gamma_rte = pb.priors.beta_theta + rte


def update_phi(pb: Problem, st: State) -> None:
    l, i, j, m = pb.nz
    xw1, _ = _weights(pb, st)
    contrib = st.rho[l, i, j, :] * xw1
    shp = np.zeros((pb.L, pb.K))
    np.add.at(shp, l, contrib)
    T = np.einsum("lijm,lm->lij", pb.R.astype(np.float64), _mean(st.gamma_shp, st.gamma_rte))
    rte = np.einsum("lijk,lij->lk", st.rho, T)
    st.phi_shp = pb.priors.alpha_lambda + shp
    st.phi_rte = pb.priors.beta_lambda + rte


def update_rho(pb: Problem, st: State) -> None:
    l, i, j, m = pb.nz
    xw1, _ = _weights(pb, st)
    l_th = _elog(st.gamma_shp, st.gamma_rte)
    l_la = _elog(st.phi_shp, st.phi_rte)
    T = np.einsum("lijm,lm->lij", pb.R.astype(np.float64), _mean(st.gamma_shp, st.gamma_rte))
    log_rho = st.logpr_rho - T[..., None] * _mean(st.phi_shp, st.phi_rte)[:, None, None, :]
    np.add.at(log_rho, (l, i, j), (l_th[l, m][:, None] + l_la[l, :]) * xw1)
    rho = np.exp(log_rho)  # no max-subtraction, as the reference (model.py:807)
    s = rho.sum(axis=3)
    pos = s > 0
    rho[pos] /= s[pos][:, None]
    st.rho = rho


def update_nu(pb: Problem, st: State) -> None:
    if not pb.mutuality:
        return
    l, i, j, m = pb.nz
    _, xw2 = _weights(pb, st)
    st.nu_shp = float(pb.priors.alpha_eta + (xw2 * st.rho[l, i, j, :]).sum())


def cavi_step(pb: Problem, st: State) -> None:
    """One sweep gamma -> phi -> rho -> nu (model.py:623-660)."""
    update_gamma(pb, st)
    update_phi(pb, st)
    update_rho(pb, st)
    update_nu(pb, st)


# --------------------------------------------------------------------------
# ELBO (model.py:948-1019, 1220-1313)
# --------------------------------------------------------------------------

def _gamma_term(pa, pb_, qa, qb):
    return gammaln(qa) - pa * np.log(qb) + (pa - qa) * psi(qa) + qa * (1.0 - pb_ / qb)


def elbo(pb: Problem, st: State) -> float:
    p = pb.priors
    l, i, j, m = pb.nz
    Rf = pb.R.astype(np.float64)
    E_th, E_la = _mean(st.gamma_shp, st.gamma_rte), _mean(st.phi_shp, st.phi_rte)
    E_nu = st.nu_shp / st.nu_rte
    XT = pb.X.transpose(0, 2, 1, 3)
    # -(sum over R of rho . (E_theta E_lambda + E_nu X^T))
    T = np.einsum("lijm,lm->lij", Rf, E_th)
    # mutuality off: the reference hands the ELBO an all-zero X^T (model.py:145, :164-170)
    Q = np.einsum("lijm,lijm->lij", Rf, XT.astype(np.float64)) if pb.mutuality else np.zeros(T.shape)
    val = -(np.einsum("lijk,lk->lij", st.rho, E_la) * T).sum() - E_nu * (st.rho.sum(axis=3) * Q).sum()
    # sum over nz(X) of x log([in R] sum_k exp(rho_k)(G_theta G_lambda_k + G_nu* xT) + eps)
    G_th = np.exp(_elog(st.gamma_shp, st.gamma_rte))
    G_la = np.exp(_elog(st.phi_shp, st.phi_rte))
    mean = G_th[l, m][:, None] * G_la[l, :] + (st.G_nu_cache * pb.xT_nz)[:, None]
    inner = (np.exp(st.rho[l, i, j, :]) * mean).sum(axis=1) * (pb.R[l, i, j, m] != 0)
    val += (pb.x_nz * np.log(inner + pb.eps)).sum()
    val += _gamma_term(p.alpha_theta, p.beta_theta, st.gamma_shp, st.gamma_rte).sum()
    val += _gamma_term(p.alpha_lambda, p.beta_lambda, st.phi_shp, st.phi_rte).sum()
    val += _gamma_term(p.alpha_eta, p.beta_eta, st.nu_shp, st.nu_rte)
    val += (st.rho * (st.logpr_rho - np.log(st.rho + pb.eps))).sum()
    val = float(val)
    if np.isnan(val):
        raise ValueError("ELBO is NaN!!!!")
    return val


# --------------------------------------------------------------------------
# fit loop (model.py:383-443, 1021-1056)
# --------------------------------------------------------------------------

@dataclass
class FitResult:
    best: State
    maxL: float
    trace: list       # (realisation, seed, iter, elbo, reached_convergence)
    next_seed: Optional[int]
    elbo_checks: list  # (realisation, iter, elbo) at every evaluation


def fit(pb: Problem, seed=None, num_realisations=1, max_iter=500, convergence_tol=0.1, decision=1,
        bias0=0.0, rho_prior=None) -> FitResult:
    maxL, best, trace, checks = -INF, None, [], []
    prng = np.random.RandomState(seed)
    for r in range(num_realisations):
        st = init_state(pb, prng, bias0=0.0 if r < 5 else (r - 4) * bias0, rho_prior=rho_prior)
        coincide, it, conv, cur = 0, 1, False, -INF
        while not conv and it <= max_iter:
            cavi_step(pb, st)
            if it == 1 or it % 10 == 0 or it == max_iter:
                old, cur = cur, elbo(pb, st)
                checks.append((r, it, cur))
                coincide = coincide + 1 if abs(cur - old) < convergence_tol else 0
            if coincide > decision:
                conv = True
            it += 1
            if (it - 1) % 10 == 0:
                trace.append((r, seed, it - 1, cur, conv))
        if maxL < cur:
            best, maxL = st, cur
        seed = next_seed(seed, prng)
        prng = np.random.RandomState(seed)
    return FitResult(best, maxL, trace, seed, checks)


def geometric_means(st: State):
    """exp(E[log .]) of theta, lambda, nu as stored in *_f (model.py:940-942)."""
    return (np.exp(_elog(st.gamma_shp, st.gamma_rte)), np.exp(_elog(st.phi_shp, st.phi_rte)),
            float(np.exp(psi(st.nu_shp) - np.log(st.nu_rte))))
