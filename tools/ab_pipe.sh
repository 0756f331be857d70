#!/bin/bash
# A/B of library builds (ab/*.so) on one box: phase times by size, the chunked device-resident schedules and the host pipeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp libeddsa_amd/libeddsa_amd.so /tmp/keep.so
for v in "$@"; do
  cp ab/$v libeddsa_amd/libeddsa_amd.so
  echo "=== $v"
  python3 tools/phase_by_size.py 2>&1 | grep mix
  python3 tools/chunked_device.py 2>&1 | grep -v amdgpu
  python3 tools/pipe_verify_sweep.py 0,0 16,20 2>&1 | grep -v amdgpu
done
cp /tmp/keep.so libeddsa_amd/libeddsa_amd.so
