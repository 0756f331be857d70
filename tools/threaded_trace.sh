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
gcc -std=c11 -O1 -pthread -Iinclude tests/c/threaded_callers.c -Llibeddsa_amd -leddsa_amd_debug -Wl,-rpath,$PWD/libeddsa_amd -o /tmp/threaded_callers
for t in 64 64 256; do /tmp/threaded_callers tests/golden/ed25519_table.bin /tmp/msgs.bin tests/golden/x25519_table.bin $t 200 48 trace | grep -v "one caller\|: ok"; done
