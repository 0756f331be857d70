#!/bin/bash
# Constant-time evidence for the comb lookup (reference discipline: lib/ed.c:346-391 scale16): the same
# instruction, LDS and bank-conflict counts whatever the secret digits are.
#   tools/ct_counters.sh <tag>  -> gpurun_out/profiles_out/<tag>_ct_counters.json   (run through gpurun)
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/ct_$TAG
rm -rf $OUT && mkdir -p $OUT $REPO/gpurun_out/profiles_out
for c in zero ones random mixed; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES \
    --output-format csv -d $OUT/$c -- python3 $REPO/tools/ct_probe.py $c > $OUT/$c.log 2>&1
done
python3 - $OUT $REPO/gpurun_out/profiles_out/${TAG}_ct_counters.json <<'PY'
import collections, csv, glob, json, sys
out = {}
for c in ("zero", "ones", "random", "mixed"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")     # "void ed::k_sign_point<4>(...)": the lanes per item stay in the name
            if k.split("<")[0].endswith("_point"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[c] = {k: {n: sorted(set(v)) for n, v in cs.items()} for k, cs in agg.items()}
kernels = sorted(out["random"])
same = {k: all(out[c].get(k) == out["random"][k] for c in out) for k in kernels}
json.dump({"identical_across_secret_classes": same, "counters": out}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print(json.dumps(same))
PY
