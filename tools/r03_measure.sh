#!/bin/bash
# The host-side measurements DESIGN.md 4 quotes, in one go (run on the GPU box): -> gpurun_out/r03_host_side.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_host_side.txt
{
echo "== tools/host_path_bench.py: 2^20 items host buffer to host buffer"
python3 tools/host_path_bench.py 2>&1 | grep -v amdgpu.ids
echo
echo "== tools/threaded_rates.sh: threads looping over the single-item functions (tests/c/threaded_callers.c), single-call latency"
tools/threaded_rates.sh 2>&1 | grep -E "threads|per call|n= "
echo
echo "== tools/verify_small.py: one pass, device-resident, ms"
python3 tools/verify_small.py 2>&1 | grep "algo 0"
echo
echo "== tools/verify_long_items.py: small passes over twelve different batches of genuine signatures"
python3 tools/verify_long_items.py 2>&1 | grep "n="
echo
echo "== tools/fixed_small.py: one pass of the fixed-base operations, device-resident, ms"
python3 tools/fixed_small.py 2>&1 | grep -v amdgpu.ids
echo
echo "== tools/x25519_small.py: one x25519 pass, device-resident, ms"
python3 tools/x25519_small.py 2>&1 | grep x25519
echo
echo "== tools/msglen_sweep.py: verify / sign against the message length, 2^18 items in HBM"
python3 tools/msglen_sweep.py 2>&1 | grep msg_len
echo
echo "== tools/chunked_device.py: what chunking alone costs (no copies)"
python3 tools/chunked_device.py 2>&1 | grep -v amdgpu.ids
echo
echo "== tools/pipe_sweep.py: chunk schedules and kernel ordering of the host pipeline"
python3 tools/pipe_sweep.py 2>&1 | grep chain
echo
echo "== tools/microbench/stage_rates.c: host copy rates"
gcc -O2 -pthread -Iinclude tools/microbench/stage_rates.c -Llibeddsa_amd -leddsa_amd -Wl,-rpath,$PWD/libeddsa_amd -o /tmp/stage_rates && /tmp/stage_rates
} > $OUT 2>&1
tail -5 $OUT
