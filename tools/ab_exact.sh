#!/bin/bash
# A/B of library builds (ab/*.so) on one box: a 2^20-item verify pass against the number of keys off the curve (tools/exact_path_time.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp libeddsa_amd/libeddsa_amd.so /tmp/keep.so
for v in "$@"; do
  cp ab/$v libeddsa_amd/libeddsa_amd.so
  echo "=== $v"
  python3 tools/exact_path_time.py 2>&1 | grep "keys"
done
cp /tmp/keep.so libeddsa_amd/libeddsa_amd.so
