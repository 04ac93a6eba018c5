"""ctypes binding of libvimure_host.so (vimure_amd/csrc/host_init.c): the one-pass, NumPy-bit-identical draw of the rho
prior (reference model.py:470-482, 536-556).  Host glue only; when the helper cannot be built the NumPy statements run."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None
_tried = False


def load():
    global _lib, _tried
    if _lib is not None or _tried:
        return _lib
    _tried = True
    try:
        path = _build.build_host()
        lib = C.CDLL(path)
        lib.vmr_host_draw_pr_rho.restype = None
        lib.vmr_host_draw_pr_rho.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        lib.vmr_host_mt_skip.restype = None
        lib.vmr_host_mt_skip.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int64]
        _lib = lib
    except Exception:   # no compiler on this host: the caller falls back to NumPy (same numbers, slower)
        _lib = None
    return _lib


_pool = None


def _threads():
    global _pool
    n = max(1, min(32, os.cpu_count() or 1))
    if _pool is None and n > 1:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="vmr-draw")
    return n, _pool


def draw_pr_rho(prng, shape, bias0, coverage, out=None, threads=None):
    """pr_rho [L,N,N,K] drawn from `prng` (np.random.RandomState; its stream advances exactly as `prng.rand(*shape)`
    would), normalised, one-hot where coverage == 0.  Returns None when the helper is unavailable.

    MT19937 can be advanced without producing numbers at ~6 G words/s (vmr_host_mt_skip: the recurrence alone), so the
    tie range is cut into blocks whose generator states are reached by skipping, and host threads draw the blocks in
    parallel (the C call releases the GIL)."""
    lib = load()
    if lib is None:
        return None
    L, N, _, K = shape
    if K > 64:
        return None
    st = prng.get_state()
    if st[0] != "MT19937":
        return None
    key0 = np.ascontiguousarray(st[1], dtype=np.uint32)
    pos0 = int(st[2])
    ties = L * N * N
    if out is None:
        out = np.empty(shape, np.float64)
    assert out.flags.c_contiguous and out.dtype == np.float64 and out.size == ties * K
    flat = out.reshape(-1)
    cov = None if coverage is None else np.ascontiguousarray(coverage, dtype=np.uint8).reshape(-1)

    def draw(key, pos, t0, t1):   # ties [t0, t1) from the generator state (key, pos) at the block's first word
        pos = C.c_int(pos)
        lib.vmr_host_draw_pr_rho(key.ctypes.data, C.byref(pos), t1 - t0, K, float(bias0),
                                 None if cov is None else cov[t0:].ctypes.data, flat[t0 * K:].ctypes.data)
        return key, pos.value

    nthr, pool = _threads()
    if threads is not None:
        nthr = max(1, min(nthr, int(threads)))
    if ties * K < (1 << 20) or pool is None or nthr < 3:
        nthr = 1
    if nthr == 1:
        key, pos = draw(key0.copy(), pos0, 0, ties)
    else:
        # One thread walks the generator from block to block WITHOUT producing numbers (the skip is a sequential chain: 64 M
        # words take 11 ms whoever does them) and publishes each block's state as it reaches it; the other threads draw the
        # blocks as their states arrive.  (Each block skipping from the start on its own cost the last one the whole chain
        # before its first number.)
        import threading
        nblk = int(min(512, max(nthr, ties * K >> 19)))
        cuts = [ties * i // nblk for i in range(nblk + 1)]
        states, ready, ends = [None] * nblk, [threading.Event() for _ in range(nblk)], [None] * nblk

        def skipper():
            key, pos = key0.copy(), C.c_int(pos0)
            for b in range(nblk):
                states[b] = (key.copy(), pos.value)
                ready[b].set()
                if b + 1 < nblk:
                    lib.vmr_host_mt_skip(key.ctypes.data, C.byref(pos), 2 * K * (cuts[b + 1] - cuts[b]))

        def worker(w, nw):
            for b in range(w, nblk, nw):
                ready[b].wait()
                ends[b] = draw(states[b][0], states[b][1], cuts[b], cuts[b + 1])

        futs = [pool.submit(skipper)] + [pool.submit(worker, w, nthr - 1) for w in range(nthr - 1)]
        for f in futs:
            f.result()
        key, pos = ends[-1]   # the last block ends where the whole draw ends
    prng.set_state(("MT19937", key, pos, st[3], st[4]))
    return out.reshape(shape)
