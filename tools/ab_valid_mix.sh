#!/bin/bash
# A/B of library builds (ab/*.so) on one box: a 2^20-item verify pass, valid signatures only and the config-2 mix, HIP-event phases
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp libeddsa_amd/libeddsa_amd.so /tmp/keep.so
for r in 1 2 3; do for v in "$@"; do
  cp ab/$v libeddsa_amd/libeddsa_amd.so
  echo "=== $v (round $r)"
  timeout 100 python3 tools/phase_by_size.py 20 2>&1 | grep "2^20"
done; done
cp /tmp/keep.so libeddsa_amd/libeddsa_amd.so
