#!/bin/bash
# A few minutes of the two threaded C programs against the real library: tools/soak.sh [threads [rounds]]
#   tests/c/host_side_stress.c   chunked ragged batches, single-item calls, trace toggling, fault hooks, concurrent shutdowns
#   tests/c/threaded_callers.c   threads looping over the eddsa.h single-item functions (the combiner)
# Every result is compared with the golden tables by the programs themselves; the exit status says whether all were right.
cd ${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-64}; R=${2:-400}
python3 - <<'PY'
import hashlib
def golden_msg(i):
    out, c = b"", 0
    while len(out) < i:
        out += hashlib.sha512(b"libeddsa-amd golden msg" + i.to_bytes(4, "little") + c.to_bytes(4, "little")).digest()
        c += 1
    return out[:i]
open("/tmp/msgs.bin", "wb").write(b"".join(golden_msg(i) for i in range(1024)))
PY
gcc -std=c11 -O1 -pthread -Iinclude tests/c/host_side_stress.c -Llibeddsa_amd -leddsa_amd_debug -Wl,-rpath,$PWD/libeddsa_amd -ldl -o /tmp/host_side_stress || exit 1
gcc -std=c11 -O1 -pthread -Iinclude tests/c/threaded_callers.c -Llibeddsa_amd -leddsa_amd_debug -Wl,-rpath,$PWD/libeddsa_amd -o /tmp/threaded_callers || exit 1
rc=0
for pass in 1 2 3; do
  s=$(date +%s.%N)
  /tmp/host_side_stress tests/golden/ed25519_table.bin /tmp/msgs.bin tests/golden/x25519_table.bin $T $R || rc=1
  e=$(date +%s.%N); echo "pass $pass: host_side_stress $T threads x $R rounds x 2 phases in $(python3 -c "print(round($e-$s,1))") s"
  /tmp/threaded_callers tests/golden/ed25519_table.bin /tmp/msgs.bin tests/golden/x25519_table.bin 256 $R 48 | grep -v "one caller" || rc=1
done
echo "soak: exit status $rc"
exit $rc
