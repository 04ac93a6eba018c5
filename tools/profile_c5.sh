#!/bin/bash
# One layer of BASELINE config 5 (N=8000, M=1000, K=3): rocprofv3 kernel statistics and HBM traffic counters (run via gpurun).
set -o pipefail
TAG=${1:-r02_c5}; N=${2:-8000}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/stress_c5.py --N $N --steps 10 > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 tools/stress_c5.py --N $N --steps 3 > $OUT/fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 tools/stress_c5.py --N $N --steps 3 > $OUT/write.log 2>&1 || exit 3
echo done
