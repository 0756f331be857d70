#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile.sh) into the small files that are
committed under profiles/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, engine kernels
only), <tag>_pmc_summary.json (per-kernel averages of every counter) and pmc_summary.json (the
latest, read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "gpurun_out", "profiles_out")
os.makedirs(dst, exist_ok=True)

def kernel_name(raw):
    """'void ed::k_verify_main_half<34>(unsigned char*, ...)' -> 'ed::k_verify_main_half' (the template arguments are the
    window count / pair bound of the pass size; a 2^20-item pass runs one instantiation)"""
    import re
    name = raw.split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    return re.sub(r"<[^>]*>$", "", name)


rows = []
for sub in ("stats_all", "stats", "stats_x25519", "stats_sign", "stats_rlc", "stats_exact"):
    for f in glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if kernel_name(r["Name"]).startswith("ed::"):
                r["Name"] = kernel_name(r["Name"])
                r["run"] = sub
                rows.append(r)
if rows:
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["run"] + [k for k in rows[0] if k != "run"])
        w.writeheader()
        w.writerows(rows)

out = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_misc", "pmc_fetch_x25519", "pmc_write_x25519",
            "pmc_fetch_sign", "pmc_write_sign", "pmc_sq_x25519", "pmc_sq_sign", "pmc_sq_exact", "pmc_fetch_exact",
            "pmc_sq_rlc", "pmc_fetch_rlc", "pmc_write_rlc"):
    for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = kernel_name(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"VGPR_Count": int(r["VGPR_Count"]), "Accum_VGPR_Count": int(r["Accum_VGPR_Count"]),
                       "SGPR_Count": int(r["SGPR_Count"]), "LDS_Block_Size": int(r["LDS_Block_Size"]),
                       "Scratch_Size": int(r["Scratch_Size"]), "Grid_Size": int(r["Grid_Size"]),
                       "Workgroup_Size": int(r["Workgroup_Size"])}
        for k, cs in agg.items():
            if sub.endswith("_exact") and "exact_lane" not in k:
                continue                       # (that workload's other kernels ran over mixed list lengths: the bench passes speak for them)
            if sub.endswith("_rlc") and "k_rlc_" not in k:
                continue                       # (tools/rlc_rate.py also times the per-item kernels: their rows come from the bench passes)
            if k.startswith("ed::"):
                # the verify workload builder also runs sign/genpub once; prefer the op's own pass
                if "_" in sub.replace("pmc_", "", 1) or not any(c in out.get(k, {}) for c in cs):
                    out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
                    out[k].update(meta[k])
# what these counters were measured on: bench.py prints them only for the tree they belong to (tools/source_hash.py)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import source_hash
out["_source"] = {"sha256": source_hash.device_source_hash(), "files": source_hash.device_source_files(), "tag": tag}
for name in (f"{tag}_pmc_summary.json", "pmc_summary.json"):
    json.dump(out, open(os.path.join(dst, name), "w"), indent=1, sort_keys=True)
for f in glob.glob(os.path.join(src, "bench_stats*.log")):
    line = [l for l in open(f) if l.startswith("{")]
    if line:
        open(os.path.join(dst, f"{tag}_{os.path.basename(f)[:-4]}.json"), "w").write(line[-1])
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read() if rows else "no stats")
m = out.get("ed::k_verify_main_half", {})
if m:
    print({k: m[k] for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "VGPR_Count", "Scratch_Size") if k in m})
