"""ctypes binding of libvimure_hip.so (C-ABI declared in include/vimure_hip.h).

There is no CPU fallback: if the shared library is missing this module raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VMR_LIB", os.path.join(HERE, "libvimure_hip.so"))  # VMR_LIB: A/B builds of the engine

VMR_OK, VMR_EINVAL, VMR_EHIP, VMR_ENAN, VMR_ESTATE = 0, -1, -2, -3, -4
STEP_GAMMA, STEP_PHI, STEP_RHO, STEP_NU = 0, 1, 2, 3
KERNEL_GAMMA_MASK, KERNEL_GAMMA_COUNTS, KERNEL_PHI, KERNEL_RHO, KERNEL_ELBO, KERNEL_FINALIZE, KERNEL_RHO_ELBO, KERNEL_RHO_NOSTORE = range(8)
READ_RHO_MAX, READ_RHO_MEAN, READ_THRESHOLD = 0, 1, 2
KERNEL_NAMES = ["gamma_mask", "gamma_counts", "phi", "rho", "elbo", "finalize", "rho_elbo", "rho_nostore"]

_dp = C.POINTER(C.c_double)
_u8p = C.c_void_p  # host or device pointer

# every symbol include/vimure_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "vmr_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             _u8p, _u8p, C.c_int, C.c_double]),
    "vmr_create_coo": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double]),
    "vmr_destroy": (None, [C.c_void_p]),
    "vmr_last_error": (C.c_char_p, [C.c_void_p]),
    "vmr_data_stats": (C.c_int, [C.c_void_p, _dp, C.c_void_p]),
    "vmr_set_priors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]),
    "vmr_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                C.c_void_p, C.c_int]),
    "vmr_step": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "vmr_elbo": (C.c_int, [C.c_void_p, _dp]),
    "vmr_fit_loop": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vmr_fit_loop_batch": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vmr_sub_step": (C.c_int, [C.c_void_p, C.c_int]),
    "vmr_sweep_local": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "vmr_commit_nu": (C.c_int, [C.c_void_p, C.c_double]),
    "vmr_sweep_local_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vmr_commit_nu_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vmr_stream": (C.c_void_p, [C.c_void_p]),
    "vmr_sample": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_int]),
    "vmr_get_state": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7),
    "vmr_get_geometric": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vmr_sync": (C.c_int, [C.c_void_p]),
    "vmr_readout": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int]),
    "vmr_snapshot": (C.c_int, [C.c_void_p]),
    "vmr_restore": (C.c_int, [C.c_void_p]),
    "vmr_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "vmr_profile_read": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int64)]),
    "vmr_kernel_bytes": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "vmr_data_format": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "vmr_mask_format": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "vmr_sweep_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "vmr_generate_y": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "vmr_generate_x": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                 C.c_uint64, C.c_int, C.c_void_p]),
    "vmr_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load the engine; raises (never falls back) when the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP engine has not been built. Run `python -m vimure_amd.build` "
            "(needs hipcc, --offload-arch=gfx950). vimure_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64; two HIP runtimes in one process cannot both own the
    # GPU.  Import torch first (when present) so that this library binds to the runtime torch uses.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("VMR_LIB_LAX") and not hasattr(lib, name):
            continue   # (A/B timing against an older experiment build picked with VMR_LIB: entry points it lacks stay unbound)
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
