#!/bin/bash
# Run on the GPU box from the repo root: per-kernel times of the general path (tools/bench_general.py), per K of the bench.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/profgen
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profgen -o g -- python3 tools/bench_general.py > gpurun_out/profgen.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/profgen/**/g_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    if "gen" in n or "fin_rho" in n:
        k = len(v) // 3
        print(n, len(v), [round(sum(v[i*k:(i+1)*k]) / max(k, 1), 1) for i in range(3)])
PY
