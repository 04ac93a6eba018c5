#!/usr/bin/env python3
"""Register / scratch table of the sweep kernels from their code objects (no GPU needed): compiles vimure_amd/csrc/sweep_sl.hip
for every K to assembly and reads the kernel descriptors' metadata.  Writes a markdown table (stdout or --out).

  python tools/kernel_resources.py [--out profiles/r03_kernel_resources.md] [--k 2 3 ...]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vimure_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def asm_for(k, tmp):
    out = os.path.join(tmp, f"sl_k{k}.s")
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                    f"-DVMR_K={k}", "--offload-device-only", "-S", "-o", out, os.path.join(CSRC, "sweep_sl.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def rows(k, text):
    out = []
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        name, blk = m.group(1), m.group(2)
        t = re.search(r"k_sweep_sl(_b)?ILi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])(?:ELb([01]))?E", name)
        if not t:
            continue
        lock = t.group(1) is not None
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
        upd, elbo, full, store, det = t.group(3) == "1", t.group(4) == "1", t.group(5) == "1", t.group(6) == "1", t.group(7) == "1"
        variant = ("update+ELBO" if elbo else "update") if upd else ("ELBO only" if elbo else "statistics only")
        if not store:
            variant += ", rho not written"
        if det:
            variant += ", deterministic"
        if lock:
            variant += " (lockstep)"
        vg = g("vgpr_count")
        alloc = (vg + 7) // 8 * 8
        out.append((k, variant, "all ones" if full else "any", vg, min(8, 512 // max(alloc, 1)), g("sgpr_count"), g("vgpr_spill_count"),
                    g("sgpr_spill_count"), g("private_segment_fixed_size")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--k", type=int, nargs="*", default=[2, 3, 4, 5, 6, 7, 8])
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max_workers=min(len(a.k), os.cpu_count() or 1)) as ex:
        texts = list(ex.map(lambda k: asm_for(k, tmp), a.k))
    table = [r for k, t in zip(a.k, texts) for r in sorted(rows(k, t), key=lambda r: (r[1], r[2]))]
    lines = ["# k_sweep_sl / k_sweep_sl_b<K, UPDATE, ELBO, ALLFULL, STORE, DET>: registers and scratch per variant (code-object metadata, gfx950)", "",
             "SGPR spills go to VGPR lanes (v_writelane), not to memory; `VGPR spills` and `scratch` are what costs.", "",
             "| K | variant | mask rows | VGPRs | waves/SIMD by VGPRs | SGPRs | VGPR spills | SGPR spills | scratch B/lane |", "|---|---|---|---|---|---|---|---|---|"]
    lines += ["| " + " | ".join(str(v) for v in r) + " |" for r in table]
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
