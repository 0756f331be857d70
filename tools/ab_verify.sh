#!/bin/bash
# A/B of library builds on one box, verify only: tools/ab_verify.sh <a.so> <b.so> ... (files under ab/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in "$@"; do
  cp ab/$v libeddsa_amd/libeddsa_amd.so
  python bench.py --op verify --steps 20 --warmup 3 --cpu-sample 4096 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), 'M/s', d['outputs_correct'], {k: round(x,3) for k,x in d['roofline']['phase_ms'].items()})"
done; done
