#!/bin/bash
# Per-kernel register / spill / scratch table from the gfx950 code-object metadata:
#   tools/kernel_resources.sh [source.hip ...]      (default: every .hip under libeddsa_amd/csrc)
cd "$(dirname "$0")/.."
SRCS=${@:-libeddsa_amd/csrc/*.hip}
for f in $SRCS; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -DEDDSA_BUILD -Iinclude -Ilibeddsa_amd/csrc -mllvm -amdgpu-dpp-combine=false \
    -S --cuda-device-only $f -o /tmp/kres.$$.s 2>/dev/null
  python3 - /tmp/kres.$$.s <<'PY'
import re, sys
t = open(sys.argv[1]).read()
print("%-32s %5s %11s %11s %8s %7s" % ("kernel", "VGPR", "vgpr_spill", "sgpr_spill", "scratch", "LDS"))
for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+)\n(.*?)\.wavefront_size", t, re.S):
    lds, name, body = m.group(1), m.group(2), m.group(3)
    if not name.startswith("_ZN2ed"): continue
    g = lambda k: re.search(k + r":\s+(\d+)", body).group(1)
    short = re.match(r"_ZN2ed\d+([a-z0-9_]+?)E", name)
    print("%-32s %5s %11s %11s %8s %7s" % (short.group(1) if short else name, g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"),
                                         g(r"\.sgpr_spill_count"), g(r"\.private_segment_fixed_size"), lds))
PY
  rm -f /tmp/kres.$$.s
done
