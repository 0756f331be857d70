#!/bin/bash
# kernel stats of the verify bench only: tools/prof_verify.sh <tag>  -> gpurun_out/prof_<tag>/verify_stats.csv
TAG=${1:-v}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 10 --warmup 2 --cpu-sample 4096 --sustained 0 --op verify ${LOG2N:+--log2n $LOG2N} > $OUT/bench.log 2>&1
cat $OUT/stats/*/*kernel_stats.csv | head -12
