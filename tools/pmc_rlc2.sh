#!/bin/bash
TAG=${1:-rlc}; R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/tools/rlc_rate.py 3 > $R/gpurun_out/$TAG/rate_pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/$TAG/pmc_tcc -- python3 $R/tools/rlc_rate.py 3 > $R/gpurun_out/$TAG/rate_pmc3.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/gpurun_out/$TAG/pmc_tcp -- python3 $R/tools/rlc_rate.py 3 > $R/gpurun_out/$TAG/rate_pmc4.log 2>&1
python3 - $R/gpurun_out/$TAG <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "rlc_bucket" in k or "verify_main" in k or "rlc_points" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
tail -3 $R/gpurun_out/$TAG/rate_pmc4.log | cut -c1-300
