"""Compile the HIP engine for gfx950 in-tree (libvimure_hip.so next to this file)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "vimure_hip.hip")
LIB = os.path.join(HERE, "libvimure_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


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


if __name__ == "__main__":
    print(build(force=True, verbose=True))
