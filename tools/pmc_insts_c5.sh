#!/bin/bash
OUT=gpurun_out/prof_sq_c5; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 tools/stress_c5.py --N 8000 --steps 4"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || echo "p1 failed"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for p in ("p1",):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, p), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:64]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen: seen.add(key); n[k] += 1
        for k in acc:
            if "k_sweep_sl" in k or "k_far" in k:
                print(p, k, n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
