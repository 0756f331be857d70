#!/bin/bash
# A/B of library builds on one box over several ops: tools/ab_ops.sh "<op> <op> ..." <a.so> <b.so> ...   (files under ab/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OPS=$1; shift
cp libeddsa_amd/libeddsa_amd.so /tmp/keep.so
for r in 1 2 3; do for v in "$@"; do for OP in $OPS; do
  cp ab/$v libeddsa_amd/libeddsa_amd.so
  python bench.py --op $OP --steps 20 --warmup 3 --cpu-sample 4096 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$OP', round(d['value']/1e6,2), 'M/s', d['outputs_correct'], round(d['roofline']['kernel_ms'],3))"
done; done; done
cp /tmp/keep.so libeddsa_amd/libeddsa_amd.so
