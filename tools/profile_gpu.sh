#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace statistics and HBM-traffic
# counters of `bench.py` on config 3.  Counters are collected in their own runs (one --pmc pass each,
# FETCH_SIZE and WRITE_SIZE do not fit one pass), never combined with sys/runtime tracing.
# Outputs land in gpurun_out/prof_<tag>/ ; tools/pmc_summary.py turns them into profiles/.
set -o pipefail
TAG=${1:-r01c}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --repeats 1 > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --repeats 1 > $OUT/fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --repeats 1 > $OUT/write.log 2>&1 || exit 3
find $OUT -name '*agent_info.csv' -delete; find $OUT/stats -name '*kernel_trace.csv' -delete
echo done
