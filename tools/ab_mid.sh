#!/bin/bash
# mid-size passes (config-2 mix, device-resident) of library builds on one box: tools/ab_mid.sh <a.so> <b.so> ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in "$@"; do cp ab/$v libeddsa_amd/libeddsa_amd.so; echo "== $v"; SIZES=19,18,17 python tools/verify_sizes.py 2>&1 | grep "n="; done; done
