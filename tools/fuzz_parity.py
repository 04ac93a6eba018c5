#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random shapes, K, mutuality, mask kinds, count ranges and engine shapes against the
coordinate-list oracle -- three sweeps with the ELBO each, state compared at the end.  `python tools/fuzz_parity.py [n] [seed] [wide|long]`.
long: many reports per tie with mirrored counts (the sweep's general body, ties sorted by the second key, level-0 rounds), two passes
and far lists forced at random."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PRI = (0.1, 0.1, 10.0, 10.0, 0.5, 1.0)


def one(case, g, wide=False, long_=False):
    """wide: the inputs beyond the specialised kernels -- K up to 70 (the reference's default K = max(X) + 1 gives such K) and counts
    beyond 11 bits / table rows beyond 2^20 (two-word entries) -- which the general kernels (csrc/sweep_gen.hip) take."""
    from oracle import cavi_coo
    from vimure_amd import CaviEngine
    L = int(g.choice([1, 1, 2, 3]))
    N = int(g.choice([5, 17, 33, 64, 65, 100, 130] if not wide else [5, 17, 33, 64, 65]))
    M = int(g.choice([1, 3, 16, 17, 40, 64, 65, 130]))
    K = int(g.choice([2, 2, 2, 3, 3, 4, 5, 8] if not wide else [2, 3, 9, 12, 16, 21, 33, 70]))
    mut = bool(g.rand() < 0.7)
    dens = float(g.choice([0.01, 0.05, 0.2, 0.5]))
    xmax = int(g.choice([1, 3, 10, 63] if not wide else [3, 40, 120, 255, 3000]))
    if long_:   # 9..40 reports per tie, many of them mirrored (levels >= 1), some ties reported by nearly everybody
        N = int(g.choice([17, 33, 64, 65, 90]))
        M = int(g.choice([64, 65, 130, 300, 640]))
        K = int(g.choice([2, 2, 3, 3, 4, 5, 8]))
        mut = bool(g.rand() < 0.85)
        dens = float(g.choice([10, 14, 19, 30, 40])) / M
        xmax = int(g.choice([1, 2, 3, 6]))
    X = ((g.rand(L, N, N, M) < dens) * g.randint(1, xmax + 1, size=(L, N, N, M))).astype(np.uint8 if xmax <= 255 else np.int32)
    if long_:
        mir = (np.transpose(X, (0, 2, 1, 3)) > 0) & (g.rand(L, N, N, M) < float(g.choice([0.0, 0.3, 0.6])))
        X = np.maximum(X, mir * g.randint(1, xmax + 1, size=(L, N, N, M))).astype(X.dtype)
        heavy = g.rand(L, N, N) < 0.01
        X[heavy] = np.maximum(X[heavy], (g.rand(int(heavy.sum()), M) < 0.9) * g.randint(1, xmax + 1, size=(int(heavy.sum()), M))).astype(X.dtype)
    mk = g.choice(["ones", "none", "random", "sparse", "self"])
    if mk == "none":
        R = None
    elif mk == "ones":
        R = np.ones_like(X)
    elif mk == "random":
        R = (g.rand(L, N, N, M) < 0.7).astype(np.uint8)
    elif mk == "sparse":
        R = (g.rand(L, N, N, M) < 0.03).astype(np.uint8)
    else:   # self-reporter-like: reporter m may report on ties that involve node m
        R = np.zeros_like(X)
        for m in range(min(M, N)):
            R[:, m, :, m] = 1
            R[:, :, m, m] = 1
    env = {}
    fmt = g.choice(["sparse", "dense", "auto"])
    if fmt != "auto":
        env["VMR_FORMAT"] = fmt
    if g.rand() < 0.3:
        env["VMR_ST_TPB"] = str(int(g.choice([64, 256, 1024])))
    if g.rand() < 0.3:
        env.update({"VMR_TWO_PASS": str(int(g.rand() < 0.5)), "VMR_YT": str(int(g.randint(0, 4))), "VMR_HC": str(int(g.randint(0, 4)))})
    if g.rand() < 0.3:
        env["VMR_TPB"] = str(int(g.choice([64, 128, 256, 512, 1024])))
    if long_:
        env["VMR_FORMAT"] = "sparse"
        env.pop("VMR_YT", None); env.pop("VMR_HC", None)
        r = g.rand()
        if r < 0.5:
            env["VMR_TWO_PASS"] = "1"
            if g.rand() < 0.3:
                env["VMR_HC"] = str(int(g.randint(1, 4)))
        elif r < 0.7:
            env["VMR_TWO_PASS"] = "0"
        else:
            env.pop("VMR_TWO_PASS", None)
        if g.rand() < 0.2:
            env["VMR_NO_LEVEL_SORT"] = "1"
        if g.rand() < 0.15:
            env["VMR_NO_LEVEL0"] = "1"
        if g.rand() < 0.15:
            env["VMR_NO_LV0R"] = "1"
    old = {k: os.environ.get(k) for k in ("VMR_FORMAT", "VMR_ST_TPB", "VMR_TWO_PASS", "VMR_YT", "VMR_HC", "VMR_TPB", "VMR_NO_LEVEL_SORT", "VMR_NO_LEVEL0",
                                          "VMR_NO_LV0R")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(env)
    desc = f"case {case}: L{L} N{N} M{M} K{K} mut={int(mut)} dens={dens} xmax={xmax} mask={mk} env={env}"
    try:
        use_coo = M <= 8192 and (xmax > 255 or (fmt != "dense" and g.rand() < 0.4))   # the coordinate-list entry point (vmr_create_coo)
        if use_coo:
            desc += " [coo]"
        try:
            if use_coo:
                sx0 = np.nonzero(X)
                eng = CaviEngine.from_coo(sx0, X[sx0], X.shape, R=None if R is None else np.nonzero(R), K=K, mutuality=mut, device=0)
            else:
                eng = CaviEngine(X, R, K=K, mutuality=mut, device=0)
        except ValueError as e:   # a refused combination: the dense tiles forced beyond 8 categories
            print(desc, "-> refused:", str(e)[:80], flush=True)
            return fmt == "dense" and K > 8 and not use_coo
        sum_x, cov = eng.data_stats()
        gs = np.random.RandomState(case)
        pr = gs.rand(L, N, N, K) + 0.05
        pr /= pr.sum(-1, keepdims=True)
        init = (0.5 + gs.rand(L, M), 0.5 + gs.rand(L, M), 1 + gs.rand(L, K), 1 + gs.rand(L, K), 0.7, 1.0 + float(sum_x), pr)
        sx = np.nonzero(X)
        c = cavi_coo.CooRef((sx, X[sx]), None if R is None else np.nonzero(R), (L, N, N, M), K, mut, PRI, *init)
        eng.set_priors(*PRI)
        eng.set_state(*init)
        ok = True
        for it in range(3):
            c.cavi_step()
            ec = c.elbo()
            try:
                e = eng.step(1, want_elbo=True)
            except ValueError as ex:   # "ELBO is NaN!!!!" (model.py:1015-1016): fine when the oracle's is NaN too
                if np.isnan(ec):
                    print(desc, f"-> both NaN at sweep {it + 1}", flush=True)
                    eng.close()
                    return True
                ok = False
                print(desc, f"-> GPU raised '{ex}' at sweep {it + 1}, oracle ELBO {ec}", flush=True)
                break
            if not abs(e - ec) <= 1e-9 * max(1.0, abs(ec)):
                ok = False
                print(desc, f"-> ELBO mismatch at sweep {it + 1}: {e} vs {ec}", flush=True)
                break
        if ok:
            st = eng.get_state()
            for name in ("gamma_shp", "gamma_rte", "phi_shp", "phi_rte"):
                a, b = st[name], getattr(c, name)
                if not np.allclose(a, b, rtol=1e-9, atol=1e-12):
                    ok = False
                    print(desc, f"-> {name} mismatch {np.max(np.abs(a - b) / (np.abs(b) + 1e-300)):.3e}", flush=True)
            if not np.allclose(st["rho"], c.rho, rtol=1e-7, atol=1e-12):
                ok = False
                print(desc, f"-> rho mismatch {np.max(np.abs(st['rho'] - c.rho)):.3e}", flush=True)
        fmt_used = eng.data_format()[0]
        eng.close()
        if ok:
            print(desc, "->", fmt_used, "ok", flush=True)
        return ok
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
    long_ = len(sys.argv) > 3 and sys.argv[3] == "long"
    g = np.random.RandomState(seed)
    t0 = time.time()
    bad = sum(0 if one(i, g, wide, long_) else 1 for i in range(n))
    print(f"{n} cases, {bad} failed, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
