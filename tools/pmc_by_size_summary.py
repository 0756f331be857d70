#!/usr/bin/env python3
"""Summarise tools/pmc_by_size.sh: per size and kernel the average duration (rocprofv3 --stats) and the counters of the
verify kernels averaged over the timed launches (the first launches of every kernel are warm-up and are dropped)."""
import csv, glob, os, sys, collections
out, sizes = sys.argv[1], sys.argv[2:]
def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ed::", "").replace("ed::", "")
            if "verify" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v[2:]) / max(1, len(v[2:])) for c, v in cs.items()} for k, cs in acc.items()}
for L in sizes:
    n = 1 << int(L)
    print(f"==== 2^{L} items ====")
    for kind in ("mix", "valid"):
        for f in glob.glob(os.path.join(out, f"stats_{L}_{kind}", "*", "*kernel_stats.csv")):
            for r in csv.DictReader(open(f)):
                if "verify" in r["Name"]:
                    print(f"  {kind:5s} {r['Name'].split('(')[0][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e6:8.3f} ms  min {float(r['MinNs']) / 1e6:8.3f}")
        for f in glob.glob(os.path.join(out, f"run_stats_{L}_{kind}.log")):
            print("  " + open(f).read().strip().splitlines()[-1])
    for grp in ("tcc", "tcc2", "sq"):
        for k, cs in sorted(counters(os.path.join(out, f"{grp}_{L}")).items()):
            line = f"  {grp:4s} {k[:44]:44s}"
            for c, v in sorted(cs.items()):
                line += f" {c}={v:.4g}"
            if "TCC_HIT_sum" in cs and cs["TCC_HIT_sum"] + cs.get("TCC_MISS_sum", 0) > 0:
                line += f" | L2 hit {cs['TCC_HIT_sum'] / (cs['TCC_HIT_sum'] + cs['TCC_MISS_sum']):.3f}"
            if "TCC_EA0_RDREQ_sum" in cs:
                line += f" | rd {cs['TCC_EA0_RDREQ_sum'] * 128 / n:.0f} B/item (at 128 B/req) wr {cs.get('TCC_EA0_WRREQ_sum', 0) * 64 / n:.0f} B/item (at 64 B/req)"
            if "GRBM_GUI_ACTIVE" in cs and "SQ_INSTS_VALU" in cs:
                line += f" | VALU-busy {cs['SQ_INSTS_VALU'] * 4 / (1024 * cs['GRBM_GUI_ACTIVE'] / 8):.3f}"
            print(line)
