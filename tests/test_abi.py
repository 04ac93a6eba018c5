"""CPU checks of the drop-in boundary: the shared library builds, loads, and exports exactly the
entry points include/vimure_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from vimure_amd import build, _lib
    build.build()          # hipcc cross-compiles gfx950 without a GPU; no-op when up to date
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vimure_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vmr_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from vimure_amd import _lib
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} not exported by libvimure_hip.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_null_handle_errors(lib):
    assert b"gfx950" in lib.vmr_version()
    # argument validation happens before any HIP call
    h = ctypes.c_void_p()
    assert lib.vmr_create(ctypes.byref(h), 0, 1, 4, 4, 1, 1, None, None, 0, 1e-12) == -1   # K < 2
    assert b"K must be" in lib.vmr_last_error(None)
    assert lib.vmr_create(ctypes.byref(h), 0, 1, 4, 4, 2, 1, None, None, 0, 1e-12) == -1   # X NULL
    assert lib.vmr_step(None, 1, None) == -1
    assert lib.vmr_elbo(None, None) == -1


def test_engine_fails_loudly_without_a_gpu():
    """No CPU fallback: constructing an engine on a box without a HIP device raises."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vimure_amd import CaviEngine
    X = np.zeros((1, 4, 4, 3), np.uint8)
    with pytest.raises(Exception) as ei:
        CaviEngine(X, None, K=2)
    assert "hip" in str(ei.value).lower() or "device" in str(ei.value).lower()


def test_missing_library_is_an_error(monkeypatch):
    from vimure_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvimure_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
