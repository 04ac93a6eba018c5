#!/usr/bin/env python3
"""profiles/<tag>_sq_counters.txt from the three --pmc passes of tools/pmc_sq.sh:  python tools/sq_summary.py gpurun_out/r03_sq profiles/r03_sq_counters.txt"""
import collections
import csv
import glob
import sys


def main(out, dst):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for p in ("p1", "p2", "p3"):
        for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if not ("k_sweep_sl" in k or "k_fin" in k):
                    continue
                acc[(k, p)][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(k, p)].add(r["Dispatch_Id"])
    lines = ["# SQ counters of bench.py's kernels (L=4 N=2000 M=200 K=2, sorted report lists), rocprofv3 --pmc, three passes (tools/pmc_sq.sh).",
             "# Values are per launch (mean over the launches of the pass).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (4 clocks) summed over waves resp. SIMDs;",
             "# SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS clocks summed over the 256 CUs; SQ_BUSY_CYCLES clocks summed over the 32 shader engines.",
             "# Template arguments of k_sweep_sl: <K, UPDATE, ELBO, ALLFULL>.", ""]
    for k in sorted({k for k, _ in acc}):
        vals = {}
        for p in ("p1", "p2", "p3"):
            if (k, p) in acc:
                for c, v in acc[(k, p)].items():
                    vals[c] = v / len(n[(k, p)])
        lines.append(k[:100])
        lines += [f"    {c:26s} {vals[c]:.0f}" for c in sorted(vals)]
        w = vals.get("SQ_WAVE_CYCLES", 0)
        if w > 0:
            lines.append(f"    -> waves waiting (any s_waitcnt) {100 * vals.get('SQ_WAIT_ANY', 0) / w:.0f} % of wave time; waiting to issue "
                         f"{100 * vals.get('SQ_WAIT_INST_ANY', 0) / w:.0f} %; VALU issuing {100 * vals.get('SQ_ACTIVE_INST_VALU', 0) / w:.0f} %")
            if vals.get("SQ_WAVES") and vals.get("SQ_BUSY_CYCLES"):
                lines.append(f"    -> mean wave lifetime {4 * w / vals['SQ_WAVES']:.0f} clocks of a kernel of {vals['SQ_BUSY_CYCLES'] / 32:.0f} clocks")
        if vals.get("SQ_LDS_IDX_ACTIVE"):
            lines.append(f"    -> LDS bank conflicts {100 * vals.get('SQ_LDS_BANK_CONFLICT', 0) / vals['SQ_LDS_IDX_ACTIVE']:.0f} % of LDS-active clocks")
        lines.append("")
    open(dst, "w").write("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
