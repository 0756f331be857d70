#!/bin/bash
# VALU-busy and LDS counters of the fixed-base point kernels: tools/pmc_sign.sh <tag> -> gpurun_out/pmc_sign_<tag>.txt
TAG=${1:-s}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_sign_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $REPO/bench.py --steps 5 --warmup 1 --cpu-sample 4096 --sustained 0 --op sign > $OUT/bench.log 2>&1
python3 - $OUT/sq <<'PY' | tee $OUT.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "sign" in k or "genpub" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in acc.items():
    m = {c: sum(v[-5:]) / len(v[-5:]) for c, v in d.items()}
    busy = 4.0 * m["SQ_INSTS_VALU"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
    print(k, "launches", len(d["ns"]), "avg ms %.3f" % (m["ns"] / 1e6), "VALU-busy %.3f" % busy, "clock GHz %.2f" % (m["GRBM_GUI_ACTIVE"] / 8 / m["ns"]),
          {c: round(v) for c, v in m.items() if c != "ns"})
PY
