"""Vectorised synthetic generators (host NumPy for small networks, torch on the GPU for large).

Same distributions as the reference's generators, without its Python triple loops
(reference synthetic.py:63-231 `_build_X`, :548-571 `StandardSBM._build_Y`, :639-667 affinity
matrix): the reference needs ~L*M*N^2 interpreter iterations and cannot produce config 3.
RNG streams differ from the reference (not bit-identical), by design.

  Y_lij ~ Poisson(u w v^T) clipped to K-1, C equal-size groups, p_in = avg_degree*C/N, p_out = 0.1 p_in,
          rescaled to N*avg_degree expected ties when `sparsify`;
  theta_lm ~ Gamma(sh_theta, sc_theta);  lambda = 0.01 for Y=0, k for Y=k (or 0.01+lambda_diff);
  M_X = theta*lambda, MM = (M_X + eta M_X^T)/(1-eta^2);
  for every unordered pair one direction ~ Poisson(MM), the other ~ Poisson(M_X + eta*first).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class SyntheticNetwork:
    X: object            # uint8 [L,N,N,M]  (numpy array or torch cuda tensor)
    R: object            # uint8 [L,N,N,M] or None (= all ones)
    Y: object            # [L,N,N] ground truth
    theta: np.ndarray    # [L,M]
    lambda_k: object     # [L,N,N]
    eta: float
    K: int


def _affinity(C, N, avg_degree, structure="assortative", a=0.1):
    p1 = avg_degree * C / N
    if structure == "assortative":
        p = p1 * a * np.ones((C, C))
        np.fill_diagonal(p, p1)
    else:
        p = p1 * np.ones((C, C))
        np.fill_diagonal(p, a * p1)
    return p


def _membership(N, C):
    size = max(1, N // C)
    return np.minimum(np.arange(N) // size, C - 1)


def self_reporter_mask(L, N, M, reporters=None):
    """R[l,i,j,m] = 1 iff m is i or j and m is a reporter (reference synthetic.py:1184-1204)."""
    assert M == N
    R = np.zeros((L, N, N, M), np.uint8)
    idx = np.arange(N) if reporters is None else np.asarray(reporters)
    R[:, idx, :, idx] = 1
    R[:, :, idx, idx] = 1
    return R


def standard_sbm(N=100, M=100, L=1, K=2, C=2, avg_degree=2.0, sparsify=True, eta=0.5, sh_theta=2.0, sc_theta=0.5,
                 flag_self_reporter=False, lambda_diff=None, theta=None, seed=0, device=None, block_bytes=2.0e9):
    """Ground truth + observed reports.  device=None -> NumPy on the host; 'cuda[:i]' -> torch on that GPU
    (X and R stay on the device as uint8 tensors, ready for CaviEngine / VimureModel.fit)."""
    if eta < 0 or eta >= 1:
        raise ValueError("The mutuality parameter has to be in [0, 1)!")
    g = np.random.RandomState(seed)
    grp = _membership(N, C)
    w = _affinity(C, N, avg_degree)
    MY = w[grp][:, grp]                      # [N,N] expected ties
    if sparsify:
        MY = MY * (float(N) * avg_degree) / MY.sum()
    if theta is None:
        theta = g.gamma(shape=sh_theta, scale=sc_theta, size=(L, M))
    if device is None:
        Y = g.poisson(np.broadcast_to(MY, (L, N, N)))
        for l in range(L):
            np.fill_diagonal(Y[l], 0)
        Y = np.minimum(Y, K - 1)
        lam = np.where(Y > 0, (0.01 + lambda_diff) if lambda_diff is not None else Y.astype(float), 0.01)
        MX = theta[:, None, None, :] * lam[..., None]
        MXt = MX.transpose(0, 2, 1, 3)
        MM = (MX + eta * MXt) / (1.0 - eta * eta)
        R = self_reporter_mask(L, N, M) if flag_self_reporter else None
        Rf = 1.0 if R is None else R.astype(float)
        first_ij = g.rand(L, N, N, M) < 0.5           # which direction of the pair (i<j) is drawn first
        iu = np.triu(np.ones((N, N), bool), 1)[None, :, :, None]
        A = g.poisson(MM * Rf)                        # candidates for the first draw, both directions
        At = A.transpose(0, 2, 1, 3)
        B = g.poisson(MX * Rf + eta * At)             # second draw, given the mirror's first
        fu = first_ij & iu                            # (i,j), i<j drawn first
        fl = (~first_ij & iu).transpose(0, 2, 1, 3)   # (j,i), i<j drawn first -> flagged at the lower triangle
        is_first = fu | fl
        is_second = is_first.transpose(0, 2, 1, 3)
        X = np.where(is_first, A, np.where(is_second, B, 0))
        if R is not None:
            X = X * (R > 0)
        X = np.minimum(X, 255).astype(np.uint8)
        return SyntheticNetwork(X, R, Y, theta, lam, eta, K)

    # On the GPU: the HIP generator kernels (csrc/generate.hip through the C-ABI: vmr_generate_y, vmr_generate_x) -- float64 rates,
    # a counter-based Philox stream, the uint8 tensor vmr_create takes written once.  torch only holds the device memory.
    Y, _ = device_sbm_y(L, N, K, C, w * ((float(N) * avg_degree) / MY.sum() if sparsify else 1.0), grp, seed, device)
    X = device_build_x(None, theta, eta, seed, Y=Y, lambda_diff=lambda_diff, flag_self_reporter=flag_self_reporter)
    import torch
    R = torch.as_tensor(self_reporter_mask(L, N, M), device=X.device) if flag_self_reporter else None
    lam = torch.where(Y > 0, torch.full(Y.shape, 0.01 + lambda_diff, device=Y.device, dtype=torch.float64) if lambda_diff is not None
                      else Y.to(torch.float64), torch.full(Y.shape, 0.01, device=Y.device, dtype=torch.float64))
    return SyntheticNetwork(X, R, Y, np.asarray(theta), lam, eta, K)


def _dev_index(device):
    import torch
    d = torch.device(device)
    if d.type != "cuda":
        raise ValueError("the device generators run on a GPU ('cuda[:i]')")
    return d, (d.index if d.index is not None else torch.cuda.current_device())


def device_sbm_y(L, N, K, C, w, grp, seed, device):
    """Ground truth of a stochastic block model drawn on the GPU (vmr_generate_y).  w [C,C]: expected ties per ordered pair of
    groups; grp [N]: group of every node.  Returns (Y uint8 [L,N,N], X uint8 [L,N,N,?] placeholder None) as torch tensors."""
    import ctypes as C_
    import torch
    from . import _lib
    lib = _lib.load()
    d, idx = _dev_index(device)
    Y = torch.empty((L, N, N), dtype=torch.uint8, device=d)
    torch.cuda.synchronize(d)
    wq = np.ascontiguousarray(w, dtype=np.float64)
    gq = np.ascontiguousarray(grp, dtype=np.int32)
    rc = lib.vmr_generate_y(idx, int(L), int(N), int(K), int(C), wq.ctypes.data, gq.ctypes.data, int(seed) & (2 ** 64 - 1), Y.data_ptr())
    if rc != 0:
        raise RuntimeError(lib.vmr_last_error(None).decode())
    return Y, None


def device_build_x(X, theta, eta, seed, Y=None, lam=None, lambda_diff=None, flag_self_reporter=False, M=None):
    """The reports X given the ground truth, drawn on the GPU (vmr_generate_x; reference `_build_X`, synthetic.py:159-231).
    Y: uint8 torch tensor [L,N,N] on the GPU (lambda = 0.01 | Y | 0.01 + lambda_diff), or lam: float64 [L,N,N] (torch GPU tensor or
    array).  X: None (allocated here) or a uint8 [L,N,N,M] GPU tensor.  Returns X."""
    import torch
    from . import _lib
    lib = _lib.load()
    src = Y if Y is not None else lam
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    L, M_ = theta.shape
    if lam is not None and not torch.is_tensor(lam):
        raise ValueError("lam must be a torch tensor on the GPU")
    d, idx = _dev_index(src.device)
    N = int(src.shape[1])
    if lam is not None:
        lam = lam.to(torch.float64).contiguous()
    if X is None:
        X = (torch.zeros if flag_self_reporter else torch.empty)((L, N, N, M_), dtype=torch.uint8, device=d)
    torch.cuda.synchronize(d)
    rc = lib.vmr_generate_x(idx, int(L), N, int(M_), Y.data_ptr() if Y is not None else None, lam.data_ptr() if lam is not None else None,
                            theta.ctypes.data, float(eta), float(lambda_diff) if lambda_diff is not None else -1.0,
                            int(seed) & (2 ** 64 - 1), int(bool(flag_self_reporter)), X.data_ptr())
    if rc != 0:
        msg = lib.vmr_last_error(None).decode()
        raise (ValueError if rc == _lib.VMR_EINVAL else RuntimeError)(msg)
    return X


# ======================================================================================================================
# The reference's generator CLASSES (synthetic.py:39-956, 1184-1254): same constructor arguments, attributes and -- in
# `exact` mode -- the same RandomState stream, so a seed gives the reference's network bit for bit (pinned by
# tests/golden/K_generators.npz).  The reference draws X with Python loops of L*M*N^2 iterations (32 s at N=500, M=50; out of
# reach at BASELINE config 3); `exact=False` draws from the same distributions in whole-array operations (NumPy, or torch
# on a GPU) -- a different stream.  Y, X, R are COO containers with the surface of sktensor's sptensor (subs, vals, shape).
# ======================================================================================================================
import math  # noqa: E402
import random as _pyrandom  # noqa: E402
import warnings as _warnings  # noqa: E402

from .tensor import SparseTensor  # noqa: E402

DEFAULT_N, DEFAULT_M, DEFAULT_L, DEFAULT_K = 100, 100, 1, 2
DEFAULT_C, DEFAULT_STRUCTURE, DEFAULT_SPARSIFY, DEFAULT_OVERLAPPING = 2, None, True, 0.0
DEFAULT_EXP_IN, DEFAULT_EXP_OUT, DEFAULT_ETA, DEFAULT_AVG_DEGREE = 2, 2.5, 0.5, 2
LAMBDA_0 = 0.01


def build_self_reporter_mask(net):
    """R[l,i,j,m] = 1 iff reporter m is the ego or the alter of the tie (reference synthetic.py:1184-1204)."""
    return self_reporter_mask(net.L, net.N, net.M).astype(float)


def build_custom_theta(net, theta_ratio=0.5, exaggeration_type="over", seed=None):
    """theta = 1 for reliable reporters, 50 ("over") or 0.5 ("under") for a random `theta_ratio` of them (synthetic.py:1207-1254)."""
    if theta_ratio < 0 or theta_ratio > 1:
        raise ValueError("theta_ratio should be in the interval [0, 1]")
    if exaggeration_type not in ["over", "under"]:
        raise ValueError("Unrecognised exaggeration_type: %s" % exaggeration_type)
    prng = np.random.RandomState(seed)
    theta = np.ones((net.L, net.M))
    n_exa = int(net.M * theta_ratio)
    chosen = prng.choice(np.arange(net.M), size=n_exa, replace=False)
    if n_exa > 0:
        theta[:, chosen] = 0.5 if exaggeration_type == "under" else 50.0
    return theta


class BaseSyntheticNetwork:
    """Ground truth Y plus `_build_X`: the observed reports X, the reporter mask R and the union / intersection baselines."""

    def __init__(self, N=DEFAULT_N, M=DEFAULT_M, L=DEFAULT_L, K=DEFAULT_K, seed=0, **kwargs):
        self.N, self.M, self.L, self.K, self.seed = N, M, L, K, seed
        self.prng = np.random.RandomState(seed)

    def _y_dense(self):
        return self.Y.toarray() if hasattr(self.Y, "toarray") else np.asarray(self.Y)

    def _build_X(self, mutuality=0.5, sh_theta=2.0, sc_theta=0.5, flag_self_reporter=True, cutoff_X=False, lambda_diff=None,
                 Q=None, seed=None, theta=None, verbose=False, exact=None, device=None):
        """Observed network X given Y (reference synthetic.py:63-352).
        exact=True: the reference's draw order, one RandomState draw at a time (its stream, bit for bit; Python loops);
        exact=False: whole-array draws from the same distributions (`device` = a torch device runs them on the GPU);
        default: exact while L*M*N^2 <= 4e6."""
        if mutuality < 0 or mutuality >= 1:
            raise ValueError("The mutuality parameter has to be in [0, 1)!")
        if seed is None:
            seed = self.seed
        prng = np.random.RandomState(seed)
        L, N, M, K = self.L, self.N, self.M, self.K
        if theta is not None:
            if type(theta) != np.ndarray or theta.shape != (L, M):
                raise ValueError("theta matrix is not valid. When using this parameter, make sure to inform a %d x %d matrix." % (L, M))
        else:
            theta = prng.gamma(shape=sh_theta, scale=sc_theta, size=(L, M))
        Y = self._y_dense()
        if lambda_diff is not None and lambda_diff <= 0:
            raise ValueError("lambda_diff is optional but when set should be higher than 0!")
        lambda_k = np.full(Y.shape, LAMBDA_0)
        for k in range(1, K):
            lambda_k[Y == k] = (LAMBDA_0 + lambda_diff) if lambda_diff is not None else k
        if cutoff_X and Q is None:
            Q = K
        if exact is None:
            exact = float(L) * M * N * N <= 4e6
        if flag_self_reporter and M != N:
            raise ValueError("flag_self_reporter needs M == N (a reporter is a node)")
        if exact:
            X, R = self._draw_exact(prng, theta, lambda_k, mutuality, flag_self_reporter, cutoff_X, Q)
        else:
            X, R = _draw_vectorised(prng, theta, lambda_k, mutuality, flag_self_reporter, device)
            if device is not None:
                X = X.cpu().numpy()
        if cutoff_X:
            X[X > Q - 1] = Q - 1
        self.X, self.R = SparseTensor.fromarray(X), SparseTensor.fromarray(R)
        self.theta, self.lambda_k, self.mutuality = theta, lambda_k, mutuality
        # baselines (synthetic.py:244-348): a tie is in the union if anybody reports it, in the intersection if everybody
        # who may report on it does
        reported = (X != 0).sum(axis=3)
        union = reported > 0
        self.X_union = SparseTensor.fromarray(union.astype(np.int8))
        inter = union & (reported == (R != 0).sum(axis=3))
        self.X_intersection = SparseTensor.fromarray(inter.astype(np.int8)) if inter.any() else None
        return self

    def _draw_exact(self, prng, theta, lambda_k, eta, flag_self_reporter, cutoff_X, Q):
        L, N, M = self.L, self.N, self.M
        MX = theta[:, None, None, :] * lambda_k[..., None]
        MM = (MX + eta * MX.transpose(0, 2, 1, 3)) / (1.0 - eta * eta)
        X = np.zeros(MM.shape, dtype=np.int64)

        def pair(l, i, j, m, Rij, Rji, det):
            """One draw of the pair (i,j)/(j,i) as seen by reporter m: a fair coin picks the direction drawn first."""
            a, b = ((i, j), (j, i)) if prng.rand(1)[0] < 0.5 else ((j, i), (i, j))
            Ra = Rij if a == (i, j) else Rji   # the reference weights BOTH means of a pair by the mask of the first direction
            first = MM[l, a[0], a[1], m] * Ra
            X[l, a[0], a[1], m] = first if det else prng.poisson(first)
            if cutoff_X and X[l, a[0], a[1], m] > Q - 1:
                X[l, a[0], a[1], m] = Q - 1
            second = MX[l, b[0], b[1], m] * Ra + eta * X[l, a[0], a[1], m]
            X[l, b[0], b[1], m] = second if det else prng.poisson(second)
        if flag_self_reporter:
            R = build_self_reporter_mask(self)
            for l in range(L):
                for m in range(M):
                    det = bool(np.allclose(theta[l, m], 1.0))   # a perfectly reliable reporter reports the means themselves
                    ii, jj = np.where(R[l, :, :, m] > 0)
                    for i, j in zip(ii.tolist(), jj.tolist()):   # both orientations of a pair are visited; the later visit wins
                        pair(l, i, j, m, R[l, i, j, m], R[l, j, i, m], det)
        else:
            R = np.ones((L, N, N, M))
            for l in range(L):
                for m in range(M):
                    for i in range(N):
                        for j in range(i + 1, N):
                            pair(l, i, j, m, 1.0, 1.0, False)
        return X, R

    def __repr__(self):
        return f"{self.__class__.__name__} (N={self.N}, M={self.M}, L={self.L}, K={self.K}, seed={self.seed})"


def _draw_vectorised(prng, theta, lambda_k, eta, flag_self_reporter, device=None):
    """X from the distributions of `_build_X` in whole-array operations (every pair once, as the all-ones branch of the
    reference; under a self-reporter mask only the reporter's own ties)."""
    L, N, _ = lambda_k.shape
    M = theta.shape[1]
    R = self_reporter_mask(L, N, M).astype(float) if flag_self_reporter else np.ones((L, N, N, M))
    if device is not None:   # the HIP generator kernel (vmr_generate_x): float64 rates, Philox stream keyed by a seed drawn from prng
        import torch
        lam = torch.as_tensor(np.ascontiguousarray(lambda_k, dtype=np.float64), device=torch.device(device))
        X = device_build_x(None, theta, eta, int(prng.randint(0, 2 ** 31 - 1)), lam=lam, flag_self_reporter=flag_self_reporter)
        return X.to(torch.int64), R
    MX = theta[:, None, None, :] * lambda_k[..., None]
    MXt = MX.transpose(0, 2, 1, 3)
    A = prng.poisson((MX + eta * MXt) / (1.0 - eta * eta) * R)
    B = prng.poisson(MX * R + eta * A.transpose(0, 2, 1, 3))
    coin = prng.rand(L, N, N, M) < 0.5
    iu = np.triu(np.ones((N, N), bool), 1)[None, :, :, None]
    first = (coin & iu) | (~coin & iu).transpose(0, 2, 1, 3)
    X = np.where(first, A, np.where(first.transpose(0, 2, 1, 3), B, 0))
    return (X * (R > 0)).astype(np.int64), R


class StandardSBM(BaseSyntheticNetwork):
    """Stochastic block model (reference synthetic.py:361-710): C equal-size groups, an assortative / disassortative
    affinity per layer, Y ~ Poisson(u w v^T) clipped to K-1."""

    def __init__(self, C=DEFAULT_C, structure=DEFAULT_STRUCTURE, avg_degree=DEFAULT_AVG_DEGREE, sparsify=DEFAULT_SPARSIFY,
                 overlapping=DEFAULT_OVERLAPPING, **kwargs):
        self._init_sbm_params(C=C, structure=structure, avg_degree=avg_degree, sparsify=sparsify, overlapping=overlapping, **kwargs)
        self._build_Y()

    def _init_sbm_params(self, C=DEFAULT_C, structure=None, avg_degree=DEFAULT_AVG_DEGREE, sparsify=False, overlapping=0.0,
                         corr=0.0, normalization=False, alpha=0.1, ag=0.1, beta=0.1, **kwargs):
        BaseSyntheticNetwork.__init__(self, **kwargs)
        self.C, self.avg_degree, self.sparsify = C, avg_degree, sparsify
        if overlapping < 0 or overlapping > 1:
            raise ValueError("The overlapping parameter has to be in [0, 1]!")
        if corr < 0 or corr > 1:
            raise ValueError("The correlation parameter corr has to be in [0, 1]!")
        self.overlapping, self.corr, self.normalization = overlapping, float(corr), bool(normalization)
        self.alpha, self.ag, self.beta = float(alpha), float(ag), float(beta)
        if structure is None:
            structure = ["assortative"] * self.L
        elif isinstance(structure, str):
            if structure not in ["assortative", "disassortative"]:
                raise ValueError("The available structures for the affinity tensor w are: assortative, disassortative!")
            structure = [structure] * self.L
        elif len(structure) != self.L:
            raise ValueError("The parameter structure should be a list of length L. "
                             "Each entry defines the structure of the corresponding layer!")
        for e in structure:
            if e not in ["assortative", "disassortative"]:
                raise ValueError("The available structures for the affinity tensor w are: assortative, disassortative.!")
        self.structure = list(structure)

    def _memberships(self):
        size = int(self.N / self.C)
        grp = np.minimum(np.arange(self.N) // max(size, 1), self.C - 1)
        u = np.zeros((self.N, self.C))
        u[np.arange(self.N), grp] = 1.0
        return u, u.copy()

    def _generate_lv(self):
        u, v = self._memberships()
        if self.overlapping > 0:   # mixed membership for a fraction of the nodes (own stream: the reference uses the global NumPy RNG here)
            n_over = int(self.N * self.overlapping)
            ind = self.prng.randint(len(u), size=n_over)
            if not self.normalization:
                u[ind] = self.prng.dirichlet(self.alpha * np.ones(self.C), n_over)
                v[ind] = self.corr * u[ind] + (1.0 - self.corr) * self.prng.dirichlet(self.alpha * np.ones(self.C), n_over)
            else:
                u[ind] = self.prng.gamma(self.ag, 1.0 / self.beta, size=(n_over, self.C))
                v[ind] = self.corr * u[ind] + (1.0 - self.corr) * self.prng.gamma(self.ag, 1.0 / self.beta, size=(n_over, self.C))
                u = u / np.where(u.sum(1, keepdims=True) == 0, 1.0, u.sum(1, keepdims=True))
            if self.normalization or self.corr > 0:
                v = v / np.where(v.sum(1, keepdims=True) == 0, 1.0, v.sum(1, keepdims=True))
        w = np.stack([_affinity(self.C, self.N, self.avg_degree, s) for s in self.structure])
        return u, v, w

    def _build_Y(self):
        self.u, self.v, self.w = self._generate_lv()
        MY = np.einsum("ijkq,akq->aij", np.einsum("ik,jq->ijkq", self.u, self.v), self.w)
        if self.sparsify:
            c = (float(self.N) * self.avg_degree) / MY.sum()
            MY *= c
            self.w *= c
        Y = self.prng.poisson(MY)
        for l in range(self.L):
            np.fill_diagonal(Y[l], 0)
        Y[Y > self.K - 1] = self.K - 1
        self.Y = SparseTensor.fromarray(Y)

    def __repr__(self):
        return (f"{self.__class__.__name__} (N={self.N}, M={self.M}, L={self.L}, C={self.C}, structure={self.structure}, "
                f"avg_degree={self.avg_degree}, sparsify={self.sparsify}, overlapping={self.overlapping})")


class DegreeCorrectedSBM(StandardSBM):
    """Degree-corrected SBM (reference synthetic.py:713-775): power-law in / out degree sequences scale the memberships."""

    def __init__(self, exp_in=DEFAULT_EXP_IN, exp_out=DEFAULT_EXP_OUT, **kwargs):
        self.exp_in, self.exp_out = exp_in, exp_out
        super().__init__(**kwargs)

    @staticmethod
    def _powerlaw_sequence(n, exponent, seed):   # networkx.utils.powerlaw_sequence: Pareto variates of the stdlib generator
        g = _pyrandom.Random(seed)
        return [g.paretovariate(exponent - 1) for _ in range(n)]

    def _generate_lv(self):
        u, v, w = super()._generate_lv()
        self.d_in = np.array([int(x) + 2 for x in self._powerlaw_sequence(self.N, self.exp_in, self.seed)])
        self.d_out = np.array([int(x) + 1 for x in self._powerlaw_sequence(self.N, self.exp_out, self.seed)])
        return u * self.d_out[:, None], v * self.d_in[:, None], w


class Multitensor(StandardSBM):
    """Community structure WITH reciprocity (reference synthetic.py:778-956, after Safdari, Contisciani & De Bacco 2021):
    (A_ij, A_ji) ~ P(A_ij) P(A_ji | A_ij), A_ij ~ Poisson(m_ij), A_ji ~ Poisson(lambda0_ji + eta A_ij)."""

    def __init__(self, eta=DEFAULT_ETA, ExpM=None, exact=None, **kwargs):
        self._init_sbm_params(**kwargs)
        if eta < 0 or eta >= 1:
            raise ValueError("The reciprocity parameter eta has to be in [0, 1)!")
        self.eta = eta
        if ExpM is None:
            self.ExpM = int(self.N * self.avg_degree / 2.0)
        else:
            self.ExpM = int(ExpM)
            self.avg_degree = 2 * self.ExpM / float(self.N)
        self._exact = (self.N <= 600) if exact is None else bool(exact)
        self._build_Y()

    def _build_Y(self):
        N, eta = self.N, self.eta
        Y = np.zeros((self.L, N, N))
        self.u, self.v, self.w = self._generate_lv()
        iu = np.triu_indices(N, 1)
        for l in range(self.L):
            M0 = np.einsum("ijkq,kq->ij", np.einsum("ik,jq->ijkq", self.u, self.v), self.w[l])
            np.fill_diagonal(M0, 0)
            if self.sparsify:
                c = (self.ExpM * (1.0 - eta)) / M0.sum()
                M0 *= c
                self.w *= c   # (every layer rescales the whole tensor, as the reference does)
            Mm = (M0 + eta * M0.T) / (1.0 - eta * eta)
            np.fill_diagonal(Mm, 0)
            A = np.zeros((N, N))
            if self._exact:   # the reference's stream: a coin, then the two Poisson draws, pair after pair
                prng = self.prng
                for i in range(N):
                    for j in range(i + 1, N):
                        if prng.rand(1)[0] < 0.5:
                            A[i, j] = prng.poisson(Mm[i, j], 1)[0]
                            A[j, i] = prng.poisson(M0[j, i] + eta * A[i, j], 1)[0]
                        else:
                            A[j, i] = prng.poisson(Mm[j, i], 1)[0]
                            A[i, j] = prng.poisson(M0[i, j] + eta * A[j, i], 1)[0]
            else:
                coin = self.prng.rand(len(iu[0])) < 0.5
                a, b = (np.where(coin, iu[0], iu[1]), np.where(coin, iu[1], iu[0]))   # (a,b) drawn first
                first = self.prng.poisson(Mm[a, b])
                A[a, b] = first
                A[b, a] = self.prng.poisson(M0[b, a] + eta * first)
            if not _weakly_connected(A):
                _warnings.warn("Multitensor has produced a network with more than one connected component. You can try increasing "
                               "avg_degree and/or running with different seeds until you get a network with just a single giant component.")
            Y[l] = A
            Y[Y > self.K - 1] = self.K - 1
        self.Y = SparseTensor.fromarray(Y)


def _weakly_connected(A):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    return connected_components(csr_matrix(A != 0), directed=True, connection="weak")[0] == 1


class PosteriorSyntheticNetwork:
    """Posterior predictive data (reference synthetic.py:964-1177): Y ~ Categorical(rho_f) of a fitted model, then X from
    theta, lambda, eta drawn from their Gamma posteriors, with the draw structure of `_build_X`."""

    def __init__(self, model, seed_Y, device=False):
        """device=True (model fitted with keep_engine=True): Y is drawn on the GPU from the rho kept there (vmr_sample, a
        Philox stream keyed by seed_Y: same distribution as the exact NumPy mode, different numbers); rho_f is not copied."""
        self.prng = np.random.default_rng(seed_Y)
        self._engine = getattr(model, "_engine", None) if device else None
        if device and self._engine is None:
            raise ValueError("device=True needs the posteriors on the GPU: fit(..., keep_engine=True)")
        self.seed_Y = seed_Y
        self.rho = None if self._engine is not None else model.rho_f
        self.theta_shp, self.theta_rte = model.gamma_shp_f, model.gamma_rte_f
        self.lambda_shp, self.lambda_rte = model.phi_shp_f, model.phi_rte_f
        self.mutuality_shp, self.mutuality_rte = model.nu_shp_f, model.nu_rte_f
        if self.rho is not None:
            self.L, self.N, self.K = self.rho.shape[0], self.rho.shape[1], self.rho.shape[3]   # (synthetic.py:1037 reads rho_f's shape)
        else:
            self.L, self.N, self.K = model.L, model.N, model.K
        self.M = self.theta_shp.shape[1]

    def build_Y(self):
        if self._engine is not None:
            Y = self._engine.sample(self.seed_Y, n_trials=1).astype(np.int64)
        else:
            Y = self.prng.multinomial(n=1, pvals=self.rho, size=(self.L, self.N, self.N)).argmax(axis=-1)
        Y[Y > self.K - 1] = self.K - 1
        self.Y = SparseTensor.fromarray(Y)

    def build_X(self, Rinput=None, flag_self_reporter=True, cutoff_X=False, Q=None, seed_X=None, verbose=True, exact=None):
        for nm, arr in (("theta_rte", self.theta_rte), ("lambda_rte", self.lambda_rte), ("mutuality_rte", self.mutuality_rte)):
            if np.any(np.asarray(arr) == 0):
                raise ValueError(f"{nm} has some zero entries!")
        if seed_X is None:
            raise ValueError("seed_X must be given")   # (the reference's default is a TypeError: np.random(90))
        self.seed_X = seed_X
        prng = np.random.RandomState(seed_X)
        L, N, M, K = self.L, self.N, self.M, self.K
        theta = prng.gamma(shape=self.theta_shp, scale=1.0 / self.theta_rte, size=(L, M))
        lambda_k = prng.gamma(shape=self.lambda_shp, scale=1.0 / self.lambda_rte, size=(L, K))
        mutuality = prng.gamma(shape=self.mutuality_shp, scale=1.0 / self.mutuality_rte, size=1)[0]
        Y = self.Y.toarray()
        lam = np.take_along_axis(lambda_k[:, None, None, :], Y[..., None].astype(np.int64), axis=3)[..., 0]
        # (absent ties, Y = 0: the reference fills lambda_k[0, 0] for every layer -- entries of its sparse Y are the non-zero ones)
        lam = np.where(Y == 0, lambda_k[0, 0], lam)
        if cutoff_X and Q is None:
            Q = K
        if exact is None:
            exact = float(L) * M * N * N <= 4e6
        helper = BaseSyntheticNetwork(N=N, M=M, L=L, K=K, seed=seed_X)
        if flag_self_reporter and Rinput is not None:
            raise NotImplementedError("a custom reporter mask: pass flag_self_reporter with the default mask, or all ones")
        if exact:
            X, R = helper._draw_exact(prng, theta, lam, mutuality, flag_self_reporter, cutoff_X, Q)
        else:
            X, R = _draw_vectorised(prng, theta, lam, mutuality, flag_self_reporter)
        if cutoff_X:
            X[X > Q - 1] = Q - 1
        self.X, self.R = SparseTensor.fromarray(X), SparseTensor.fromarray(R)
        self.theta, self.lambda_k, self.lambda_k_auxiliary, self.mutuality = theta, lambda_k, lam, mutuality
        reported = (X != 0).sum(axis=3)
        self.X_union = SparseTensor.fromarray((reported > 0).astype(np.int8))
        self.X_intersection = SparseTensor.fromarray((reported == 2).astype(np.int8))   # "both ends report it" (synthetic.py:1158-1161)
        return self
