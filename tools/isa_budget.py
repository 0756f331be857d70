#!/usr/bin/env python3
"""Static instruction budget of a kernel from its gfx950 assembly: the loops the compiler kept (body length and instruction mix) and
the straight-line rest.  No GPU needed.   tools/isa_budget.py <kernel-name-substring> [source.hip]
Used for profiles/r05_x25519_step_budget.txt and profiles/r05_prepare_halve_budget.txt (the trip counts are the source's)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "libeddsa_amd", "csrc", "kernels.hip")
asm = os.path.join(tempfile.gettempdir(), "isa_budget_" + os.path.basename(src) + ".s")
if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(os.path.join(os.path.dirname(src), f)) for f in os.listdir(os.path.dirname(src))):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-DEDDSA_BUILD", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.dirname(src), "-mllvm", "-amdgpu-dpp-combine=false", "-S", "--cuda-device-only", src, "-o", asm], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN2ed\d+" + ".*" + re.escape(want) + r".*:\s", l)]
for start in starts:
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel " + name in lines[i])
    body = lines[start:end]
    ins = lambda seg: [l.split()[0] for l in (x.strip() for x in seg) if l and not l.startswith((";", ".")) and not l.endswith(":")]  # noqa: E731
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = sorted({(labels[m.group(1)], i) for i, l in enumerate(body) for m in [re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l)]
                    if m and labels.get(m.group(1), 1 << 30) < i})
    allv = ins(body)
    print(f"== {name}: {len(allv)} instructions in the code object, {sum(1 for o in allv if o.startswith('v_'))} of them VALU")
    inner = [(a, b) for a, b in loops if not any(c > a and d < b for c, d in loops)]
    for a, b in loops:
        ops = ins(body[a:b + 1])
        c = collections.Counter(ops)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        kind = "innermost" if (a, b) in inner else "outer"
        print(f"  loop at +{a:5d} .. +{b:5d} ({kind}): {len(ops):5d} instructions, {valu:5d} VALU: " +
              ", ".join(f"{k} {v}" for k, v in c.most_common(9)))
