#!/bin/bash
# The device source (lanes.h and the layers under it) compiled for the host with -DED_HOST_CHECK and
# AddressSanitizer + UndefinedBehaviorSanitizer, driven by the CPU suite's host-check tests.
# (GPU sanitizers are not available on the pool; this is the CPU-side equivalent.)
set -e
cd "$(dirname "$0")/.."
SAN=/tmp/libhostcheck_san.so
g++ -std=c++17 -O1 -g -fPIC -shared -DED_HOST_CHECK -Wno-unknown-pragmas -fsanitize=address,undefined \
    -fno-sanitize-recover=undefined -Ilibeddsa_amd/csrc tests/host_check/host_check.cpp -o $SAN
KEEP=$(mktemp)
[ -f tests/host_check/libhostcheck.so ] && cp tests/host_check/libhostcheck.so $KEEP
cp $SAN tests/host_check/libhostcheck.so
trap '[ -s $KEEP ] && cp $KEEP tests/host_check/libhostcheck.so || rm -f tests/host_check/libhostcheck.so' EXIT
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_device_source_on_host.py -x -q
# ... and the product's HOST side (eddsa_amd.c + host_pipe.c, unchanged) against the fake HIP runtime / fake RCCL of
# tests/fake_hip/, under -fsanitize=thread and -fsanitize=address,undefined, with 2, 3 and 8 devices (incl. the walk that
# fails every runtime and RCCL call in turn: tests/c/host_fault_walk.c)
python -m pytest tests/test_host_side_sanitized.py -x -q
