#!/bin/bash
# A/B timing of engine builds / runtime shapes on bench.py's config (development aid; run via gpurun from the repo root).
#   tools/ab_bench.sh OUTDIR "name|ENV=..,ENV=.." ...      (VMR_LIB=tools/_bin/x.so picks a build)
OUT=$1; shift
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
    timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-converge ${BENCH_ARGS} > $OUT/$name.log 2>&1
    echo "rc=$?" >> $OUT/$name.log )
  python3 - "$OUT/$name.log" "$name" <<'PY'
import json, sys
s = open(sys.argv[1]).read()
try:
    j = json.loads(s[s.index('{"metric'):].splitlines()[0])
    k = j["kernels"]
    print("%-14s %8.1f it/s  rho %.4f ms  rho_elbo %.4f ms  fin %.4f ms  stats %.4f" % (
        sys.argv[2], j["value"], k["rho"]["avg_ms"], k["rho_elbo"]["avg_ms"], k["finalize"]["avg_ms"], k["gamma_counts"]["avg_ms"]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", s[-400:].replace("\n", " | "), flush=True)
PY
done
