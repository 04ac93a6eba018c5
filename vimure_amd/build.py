"""Compile the HIP engine for gfx950 in-tree (libvimure_hip.so next to this file)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "vimure_hip.hip")
LIB = os.path.join(HERE, "libvimure_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOST_SRC = os.path.join(HERE, "csrc", "host_init.c")
HOST_LIB = os.path.join(HERE, "libvimure_host.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = [SRC, os.path.join(ROOT, "include", "vimure_hip.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-std=c++17",
           "-I" + os.path.join(ROOT, "include"), "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    """The host-side helper (initial-state draw, plain C): gcc, no FMA contraction (NumPy rounds twice)."""
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= os.path.getmtime(HOST_SRC):
        return HOST_LIB
    cmd = [os.environ.get("CC", "gcc"), "-O3", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-o", HOST_LIB, HOST_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return HOST_LIB


if __name__ == "__main__":
    print(build_host(force=True, verbose=True))
    print(build(force=True, verbose=True))
