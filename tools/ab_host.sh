#!/bin/bash
# A/B of library builds (ab/*.so) on one box, host pipeline only, builds alternating: tools/ab_host.sh a.so b.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp libeddsa_amd/libeddsa_amd.so /tmp/keep.so
for r in 1 2 3; do for v in "$@"; do
  cp ab/$v libeddsa_amd/libeddsa_amd.so
  echo "=== $v (round $r)"
  timeout 100 python3 tools/pipe_verify_sweep.py 0,0 2>&1 | grep "host to host"
done; done
cp /tmp/keep.so libeddsa_amd/libeddsa_amd.so
