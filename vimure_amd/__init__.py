"""vimure_amd -- MI355X-native CAVI fitting engine for the VIMuRe latent-network model.

Host API mirrors latentnetworks/vimure (`VimureModel.fit`, `get_inferred_model`, ...);
the coordinate-ascent sweeps and the ELBO run in hand-written HIP kernels for gfx950
behind the C-ABI of include/vimure_hip.h.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .engine import CaviEngine, EngineError  # noqa: F401

__all__ = ["CaviEngine", "EngineError", "VimureModel"]


def __getattr__(name):
    if name == "VimureModel":
        from .model import VimureModel
        return VimureModel
    if name in ("model", "synthetic", "tensor"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
