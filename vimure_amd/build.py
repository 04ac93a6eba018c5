"""Compile the HIP engine for gfx950 in-tree (libvimure_hip.so next to this file).

The library is a handful of translation units (vimure_amd/csrc): objects are compiled in parallel into csrc/_obj and only
those whose source or headers changed are rebuilt; the sweep kernel is one object per number of categories K (-DVMR_K).
`VMR_DEV=1` builds K = 2 only (fast iteration on a kernel)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.environ.get("VMR_LIB_OUT", os.path.join(HERE, "libvimure_hip.so"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOST_SRC = os.path.join(CSRC, "host_init.c")
HOST_LIB = os.path.join(HERE, "libvimure_host.so")
HEADERS = [os.path.join(CSRC, "vmr_internal.h"), os.path.join(CSRC, "sweep_sl.h"), os.path.join(CSRC, "sweep_gen.h"),
           os.path.join(ROOT, "include", "vimure_hip.h")]
KS = (2, 3, 4, 5, 6, 7, 8)
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def units(dev=False):
    """(object name, source, extra flags) of every translation unit."""
    extra = ["-DVMR_DEV"] if dev else []
    u = [("vimure_hip", os.path.join(CSRC, "vimure_hip.hip"), extra),
         ("sorted_lists", os.path.join(CSRC, "sorted_lists.hip"), extra),
         ("sweep_gen", os.path.join(CSRC, "sweep_gen.hip"), extra),   # the general kernels: any K, wide entries
         ("generate", os.path.join(CSRC, "generate.hip"), extra)]     # the synthetic generators on the device
    dev_ks = tuple(int(k) for k in os.environ.get("VMR_DEV_KS", "2").split(","))   # (VMR_DEV_KS=2,3: also the K = 3 sweep kernels)
    for k in (dev_ks if dev else KS):
        u.append((f"sweep_sl_k{k}", os.path.join(CSRC, "sweep_sl.hip"), extra + [f"-DVMR_K={k}"]))
    return u


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + HEADERS)


def needs_build() -> bool:
    dev = bool(os.environ.get("VMR_DEV"))
    tag = os.path.join(OBJ, "dev" if dev else "full")
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in [u[1] for u in units(dev)] + HEADERS) or not os.path.exists(tag)


def build(force: bool = False, verbose: bool = False) -> str:
    """VMR_CXXFLAGS adds compiler flags (kernel experiments: -DSL_WPE=5 ...), VMR_TAG keeps such a build's objects apart
    (csrc/_obj/<tag>), VMR_LIB_OUT names the library -- tools/ab_bench.sh picks builds with VMR_LIB."""
    tagged = bool(os.environ.get("VMR_TAG"))
    if not force and not tagged and not needs_build():
        return LIB
    dev = bool(os.environ.get("VMR_DEV"))
    odir = os.path.join(OBJ, os.environ.get("VMR_TAG") or ("dev" if dev else "full"))
    more = os.environ.get("VMR_CXXFLAGS", "").split()
    os.makedirs(odir, exist_ok=True)
    todo, objs = [], []
    for name, src, extra in units(dev):
        obj = os.path.join(odir, name + ".o")
        objs.append(obj)
        if force or _stale(obj, src):
            todo.append([HIPCC] + FLAGS + extra + more + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    workers = max(1, min(len(todo), int(os.environ.get("VMR_BUILD_JOBS", os.cpu_count() or 1))))
    if todo:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(run, todo))
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
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
    print(build(force="--force" in __import__("sys").argv, verbose=True))
