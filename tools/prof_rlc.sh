#!/bin/bash
# rocprofv3 kernel stats of tools/rlc_rate.py (run through gpurun): tools/prof_rlc.sh <tag>
TAG=${1:-rlc}; R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$TAG; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/prof -- python3 $R/tools/rlc_rate.py 5 > $R/gpurun_out/$TAG/rate_profiled.log 2>&1
python3 - $R/gpurun_out/$TAG <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if "rlc" in n: print("%-28s calls %3s avg %8.3f ms  min %8.3f" % (n, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6))
for l in open(sys.argv[1] + "/rate_profiled.log"):
    if l.startswith("{"):
        d = json.loads(l); print("per-item %.1f M/s, rlc %.1f M/s (%.2f ms), speedup %.2f" % (d["per_item"]["verifies_per_s"] / 1e6, d["rlc"]["verifies_per_s"] / 1e6, d["rlc"]["ms"], d["speedup"]))
PY
