#!/bin/bash
# A/B builds of the library on the same box, alternating:   tools/ab.sh <rounds> '<command>' ab/A.so ab/B.so ...
# Each variant is loaded through EDDSA_AMD_LIBRARY (libeddsa_amd/api.py: library_path); the product library is never touched.
set -u
R=$1; CMD=$2; shift 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in $(seq "$R"); do for v in "$@"; do
  echo "== $v (round $r)"
  EDDSA_AMD_LIBRARY="$PWD/$v" bash -c "$CMD" 2>&1 | tail -${AB_TAIL:-3}
done; done
