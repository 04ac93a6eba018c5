"""ctypes binding of libvimure_host.so (vimure_amd/csrc/host_init.c): the one-pass, NumPy-bit-identical draw of the rho
prior (reference model.py:470-482, 536-556).  Host glue only; when the helper cannot be built the NumPy statements run."""
import ctypes as C
import os
import threading

import numpy as np

from . import build as _build

_lib = None
_tried = False
_load_lock = threading.Lock()


def load():
    """Built and bound once; callers that arrive while another thread is still building wait for it (batch.py creates
    engines from several threads at once)."""
    global _lib, _tried
    if _lib is not None or _tried:
        return _lib
    with _load_lock:
        if _lib is not None or _tried:
            return _lib
        _lib = _load_locked()
        _tried = True
    return _lib


def _load_locked():
    try:
        path = _build.build_host()
        lib = C.CDLL(path)
        lib.vmr_host_draw_pr_rho.restype = None
        lib.vmr_host_draw_pr_rho.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        lib.vmr_host_mt_skip.restype = None
        lib.vmr_host_mt_skip.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int64]
        return lib
    except Exception:   # no compiler on this host: the caller falls back to NumPy (same numbers, slower)
        return None


_pool = None
max_threads = None   # a cap on the threads of ONE draw (the lockstep batch driver draws many states side by side: 1 each)


def _threads():
    global _pool
    n = max(1, min(32, os.cpu_count() or 1))
    if max_threads is not None:
        n = max(1, min(n, int(max_threads)))
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
    if K >= 8:   # NumPy's sum(axis=-1) switches to its 8-accumulator pairwise order at 8 terms; the C pass adds left to right
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


def mt_skip(prng, n_doubles):
    """Advance `prng` (np.random.RandomState) past n_doubles values of `random_sample` without producing them (two 32-bit
    outputs each; vmr_host_mt_skip).  Returns False when the helper is unavailable (the caller then draws and drops)."""
    lib = load()
    if lib is None or n_doubles < 0:
        return False
    st = prng.get_state()
    if st[0] != "MT19937":
        return False
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int(int(st[2]))
    lib.vmr_host_mt_skip(key.ctypes.data, C.byref(pos), 2 * int(n_doubles))
    prng.set_state(("MT19937", key, pos.value, st[3], st[4]))
    return True


def draw_pr_rho_layers(prng, L_total, N, K, bias0, layers, coverage_local):
    """The rows `layers` (sorted) of the pr_rho that `draw_pr_rho(prng, (L_total, N, N, K), ...)` would return, leaving `prng`
    where the full draw leaves it: the generator skips the other layers' numbers instead of producing them (a rank of a
    layer-sharded fit owns 1 of 8 layers of 192 M doubles each).  coverage_local: [len(layers), N, N].  None when the helper is
    unavailable."""
    if load() is None or K >= 8:
        return None
    out = np.empty((len(layers), N, N, K), np.float64)
    per = N * N * K
    cursor = 0
    for q, l in enumerate(layers):
        if l < cursor:
            raise ValueError("layers must be sorted and distinct")
        if not mt_skip(prng, (l - cursor) * per):
            return None
        cov = None if coverage_local is None else coverage_local[q:q + 1]
        if draw_pr_rho(prng, (1, N, N, K), bias0, cov, out=out[q:q + 1]) is None:
            return None
        cursor = l + 1
    if not mt_skip(prng, (L_total - cursor) * per):
        return None
    return out
