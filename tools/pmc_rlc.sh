#!/bin/bash
# PMC counters of the RLC kernels (separate pass, no tracing): tools/pmc_rlc.sh <tag>
TAG=${1:-rlc}; R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc -- python3 $R/tools/rlc_rate.py 3 > $R/gpurun_out/$TAG/rate_pmc.log 2>&1
python3 - $R/gpurun_out/$TAG <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "rlc" in k or "verify_main" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
