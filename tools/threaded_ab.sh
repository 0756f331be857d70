#!/bin/bash
# threaded single-item callers against two builds of the library: tools/threaded_ab.sh ab/C.so ab/D.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<'PY'
import hashlib
def golden_msg(i):
    out, c = b"", 0
    while len(out) < i:
        out += hashlib.sha512(b"libeddsa-amd golden msg" + i.to_bytes(4, "little") + c.to_bytes(4, "little")).digest()
        c += 1
    return out[:i]
open("/tmp/msgs.bin","wb").write(b"".join(golden_msg(i) for i in range(1024)))
PY
# the program binds libeddsa_amd_debug.so (that build's SONAME; variants are debug builds): each gets a directory of its own with that name in it - nothing in the tree is overwritten
gcc -std=c11 -O1 -pthread -Iinclude tests/c/threaded_callers.c -Llibeddsa_amd -leddsa_amd_debug -o /tmp/threaded_callers
for rep in 1 2; do for so in "$@"; do
  d=$(mktemp -d); cp "$so" "$d/libeddsa_amd_debug.so"; echo "== $so"
  for t in 8 32 64 128 256 512; do LD_LIBRARY_PATH="$d" /tmp/threaded_callers tests/golden/ed25519_table.bin /tmp/msgs.bin tests/golden/x25519_table.bin $t 200 48 | grep -v "one caller\|: ok"; done
  rm -rf "$d"
done; done
