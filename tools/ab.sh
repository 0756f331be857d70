#!/bin/bash
# A/B two builds of the library on the same box: tools/ab.sh <op> [rounds]  (ab/A.so, ab/B.so)
OP=${1:-verify}; R=${2:-3}
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq $R); do for v in A B; do
  cp ab/$v.so libeddsa_amd/libeddsa_amd.so
  python bench.py --op $OP --steps 20 --warmup 3 --cpu-sample 4096 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$OP', round(d['value']/1e6,2), 'M/s', d['outputs_correct'], d['roofline'].get('phase_ms'))"
done; done
