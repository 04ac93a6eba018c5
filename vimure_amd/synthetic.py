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

    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    th = torch.as_tensor(theta, device=dev, dtype=torch.float32)
    MYt = torch.as_tensor(MY, device=dev, dtype=torch.float32)
    X = torch.zeros((L, N, N, M), dtype=torch.uint8, device=dev)
    Ys, lams = [], []
    # row blocks small enough that a [B,B,M] float32 temporary stays near 2 GB (config 5: N=8000, M=1000 -> B=724)
    B = int(min(N, max(1, int((block_bytes / (4.0 * M)) ** 0.5))))
    starts = list(range(0, N, B))
    for l in range(L):
        Y = torch.poisson(MYt, generator=gen)
        Y.fill_diagonal_(0)
        Y.clamp_(max=K - 1)
        lam = torch.where(Y > 0, torch.full_like(Y, 0.01 + lambda_diff) if lambda_diff is not None else Y,
                          torch.full_like(Y, 0.01))
        for bi, i0 in enumerate(starts):
            i1 = min(N, i0 + B)
            for j0 in starts[bi:]:
                j1 = min(N, j0 + B)
                MXa = lam[i0:i1, j0:j1, None] * th[l][None, None, :]            # block (I,J)
                MXb = lam[j0:j1, i0:i1, None] * th[l][None, None, :]            # block (J,I)
                Aa = torch.poisson((MXa + eta * MXb.transpose(0, 1)) / (1.0 - eta * eta), generator=gen)
                # on a diagonal block (J,I) IS (I,J): one draw serves both directions
                Ab = Aa if i0 == j0 else torch.poisson((MXb + eta * MXa.transpose(0, 1)) / (1.0 - eta * eta), generator=gen)
                Ba = torch.poisson(MXa + eta * Ab.transpose(0, 1), generator=gen)   # second draw given the mirror's first
                Bb = torch.poisson(MXb + eta * Aa.transpose(0, 1), generator=gen)
                coin = torch.rand(MXa.shape, device=dev, generator=gen) < 0.5     # (i,j) drawn first, else (j,i)
                gi = torch.arange(i0, i1, device=dev)[:, None, None]
                gj = torch.arange(j0, j1, device=dev)[None, :, None]
                up = gi < gj                                                      # i < j: this element leads its pair
                lo = gi > gj
                if i0 == j0:   # diagonal block: both triangles live in the same block
                    coin_t = coin.transpose(0, 1)
                    Xa = torch.where(up & coin, Aa, torch.where(up & ~coin, Ba,
                         torch.where(lo & coin_t, Ba, torch.where(lo & ~coin_t, Aa, torch.zeros_like(Aa)))))
                    X[l, i0:i1, j0:j1] = Xa.clamp_(max=255).to(torch.uint8)
                else:          # every (i,j) of the block has i < j
                    X[l, i0:i1, j0:j1] = torch.where(coin, Aa, Ba).clamp_(max=255).to(torch.uint8)
                    X[l, j0:j1, i0:i1] = torch.where(coin.transpose(0, 1), Bb, Ab).clamp_(max=255).to(torch.uint8)
                del MXa, MXb, Aa, Ab, Ba, Bb, coin
        Ys.append(Y.to(torch.uint8))
        lams.append(lam)
    R = None
    if flag_self_reporter:
        R = torch.as_tensor(self_reporter_mask(L, N, M), device=dev)
        X = X * R
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return SyntheticNetwork(X, R, torch.stack(Ys), np.asarray(theta), torch.stack(lams), eta, K)
