#!/bin/bash
# VALU instruction counts of the verify kernels: tools/pmc_verify.sh <tag>
TAG=${1:-v}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 4096 --op verify > $OUT/bench_pmc.log 2>&1
python3 - $OUT/pmc <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "verify" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "VGPR", "n=%d" % len(next(iter(d.values()))))
PY
