#!/bin/bash
# small-pass timings of library builds on one box: tools/ab_small.sh <a.so> <b.so> ... (files under ab/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do cp ab/$v libeddsa_amd/libeddsa_amd.so; echo "== $v"; python tools/verify_small.py 2>&1 | grep "algo 0"; done
