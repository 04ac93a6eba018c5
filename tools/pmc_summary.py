#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (made by tools/profile_gpu.sh) into profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (our kernels only)
  profiles/<tag>_hbm_traffic.csv    per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
  profiles/pmc_traffic.json         {config: {kernel class: bytes per launch}} read by bench.py
Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced streaming read, so the
read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import collections
import csv
import glob as _glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class glob:   # newest match first (a profile directory may hold earlier runs)
    @staticmethod
    def glob(pat):
        return sorted(_glob.glob(pat), key=os.path.getmtime, reverse=True)
CLASS = [("k_gamma_mask", "gamma_mask"), ("k_gamma_counts", "gamma_counts"), ("k_phi", "phi"),
         ("true, true, true", "rho_elbo"), ("false, true, true", "rho_elbo"), ("k_rho", "rho")]


def kclass(name):
    for key, cls in CLASS:
        if key in name:
            if cls == "rho" and ", false, true," in name.replace("true, false, true", ", false, true,"):
                pass
            return cls
    return None


def per_kernel(path, counter):
    rows = list(csv.DictReader(open(path)))
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    st = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
    if st:
        rows = list(csv.DictReader(open(st[0])))
        keep = [r for r in rows if r["Name"].lstrip("void ").startswith("k_")]
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
        print("kernel stats:", len(keep), "kernels")
    fe = glob.glob(os.path.join(src, "fetch", "*", "*counter_collection.csv"))
    wr = glob.glob(os.path.join(src, "write", "*", "*counter_collection.csv"))
    if fe and wr:
        F, W = per_kernel(fe[0], "FETCH_SIZE"), per_kernel(wr[0], "WRITE_SIZE")
        traffic = {}
        with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "FETCH_SIZE_KiB_raw", "WRITE_SIZE_KiB_raw", "read_bytes_corrected_x2", "write_bytes",
                        "hbm_bytes_per_launch"])
            for k in sorted(F):
                if "k_" not in k:
                    continue
                rd, wb = 2.0 * F[k] * 1024.0, W.get(k, 0.0) * 1024.0
                w.writerow([k, F[k], W.get(k, 0.0), rd, wb, rd + wb])
                name = k
                cls = None
                if "k_gamma_mask" in name: cls = "gamma_mask"
                elif "k_gamma_counts" in name: cls = "gamma_counts"
                elif "k_phi" in name: cls = "phi"
                elif "k_rho" in name or "k_sweep_sl" in name:   # k_rho / k_rho_sp<K, MUT, UPDATE, ELBO..>, k_sweep_sl<K, UPDATE, ELBO, ALLFULL>
                    flags = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
                    o = 1 if "k_sweep_sl" in name else 2
                    upd, elbo = flags[o] == "true", flags[o + 1] == "true"
                    cls = "rho_elbo" if (upd and elbo) else ("rho" if upd else ("elbo" if elbo else "gamma_counts"))
                    if cls == "rho" and "k_sweep_sl" in name and len(flags) > o + 3 and flags[o + 3] == "false":
                        cls = "rho_nostore"   # <K, UPDATE, ELBO, ALLFULL, STORE, DET>: rho used, not written
                if cls:
                    traffic[cls] = rd + wb
        pj = os.path.join(dst, "pmc_traffic.json")
        cur = json.load(open(pj)) if os.path.exists(pj) else {}
        cur[sys.argv[2] if len(sys.argv) > 2 else "c3"] = traffic
        json.dump(cur, open(pj, "w"), indent=1, sort_keys=True)
        print("traffic:", {k: f"{v/1e9:.3f} GB" for k, v in traffic.items()})


if __name__ == "__main__":
    main()
