#!/bin/bash
# Instruction-fetch counters of one verify pass of 2^20 items and of the sign / x25519 passes, per kernel:
#   tools/pmc_ifetch.sh   -> gpurun_out/pmc_ifetch/summary.txt
# (is a kernel whose straight-line code is larger than the 64 KB instruction cache held up by its fetches?)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_ifetch
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE --output-format csv -d $OUT/ic_v -- python3 $REPO/tools/verify_pass.py 20 5 mix > $OUT/run_ic_v.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/if_v -- python3 $REPO/tools/verify_pass.py 20 5 mix > $OUT/run_if_v.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE --output-format csv -d $OUT/ic_s -- python3 $REPO/bench.py --op sign --steps 5 --warmup 2 --cpu-sample 4096 > $OUT/run_ic_s.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/if_s -- python3 $REPO/bench.py --op sign --steps 5 --warmup 2 --cpu-sample 4096 > $OUT/run_if_s.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE --output-format csv -d $OUT/ic_x -- python3 $REPO/bench.py --op x25519 --steps 5 --warmup 2 --cpu-sample 4096 > $OUT/run_ic_x.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/if_x -- python3 $REPO/bench.py --op x25519 --steps 5 --warmup 2 --cpu-sample 4096 > $OUT/run_if_x.log 2>&1
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ed::", "").replace("ed::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-28s %10s %8s %10s %12s %12s %10s" % ("kernel", "I$ req", "hit", "miss/wave", "ifetch/wave", "avg level", "VALU-busy"))
for k, cs in sorted(acc.items()):
    m = {c: sum(v[1:]) / max(1, len(v[1:])) for c, v in cs.items()}
    if m.get("SQ_WAVES", 0) < 1000: continue
    req, hit, miss = m.get("SQC_ICACHE_REQ", 0), m.get("SQC_ICACHE_HITS", 0), m.get("SQC_ICACHE_MISSES", 0)
    waves = m["SQ_WAVES"]
    busy = m.get("SQ_INSTS_VALU", 0) * 4 / (1024 * m["GRBM_GUI_ACTIVE"] / 8) if m.get("GRBM_GUI_ACTIVE") else 0
    lvl = m.get("SQ_IFETCH_LEVEL", 0) / m["SQ_IFETCH"] if m.get("SQ_IFETCH") else 0
    print("%-28s %10.3g %8.3f %10.1f %12.1f %12.2f %10.3f" % (k[:28], req, hit / req if req else 0, miss / waves, m.get("SQ_IFETCH", 0) / waves, lvl, busy))
PY
