#!/bin/bash
# SQ instruction / cycle counters of bench.py's kernels (development aid; run via gpurun from the repo root).
#   tools/pmc_sq.sh [OUTDIR]     (env BENCH_ARGS adds bench.py flags; VMR_* select builds / shapes)
set -o pipefail
OUT=${1:-gpurun_out/prof_sq}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-converge --no-extra --repeats 1 $BENCH_ARGS"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || echo "p1 failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || echo "p2 failed"
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/p3 -- $B > $OUT/p3.log 2>&1 || echo "p3 failed"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for p in ("p1", "p2", "p3"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, p), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:64]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen: seen.add(key); n[k] += 1
        for k in acc:
            if "k_rho_sp" in k or "k_fin" in k or "k_sweep_sl" in k:
                print(p, k, n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
