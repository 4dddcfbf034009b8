#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_bench.sh into what is kept
under profiles/: the kernel-stats table as is, and per-dispatch counter rows of
OUR kernels only (the raw collection also lists every rocclr copy kernel).

usage: tools/summarize_pmc.py gpurun_out/<dir> profiles/<prefix> [kernel-substring ...]
"""
import collections
import csv
import json
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
keys = sys.argv[3:] or ["ssvs_", "kalman", "atb_mfma", "xtx_mfma"]
shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), prefix + "_kernel_stats.csv")
summary = {"source": src, "kernels": {}}
rows_out = []
for name in sorted(os.listdir(src)):
    path = os.path.join(src, name, "pmc_counter_collection.csv")
    if not (name.startswith("pmc_") and os.path.exists(path)):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"]
        if not any(k in kn for k in keys):
            continue
        short = kn.split("(")[0].replace("void ", "").replace("boom_amd::", "").replace("(anonymous namespace)::", "")
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[short] = dict(grid=r["Grid_Size"], workgroup=r["Workgroup_Size"], lds=r["LDS_Block_Size"],
                           scratch=r["Scratch_Size"], vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"])
        rows_out.append([name, r["Dispatch_Id"], short, r["Counter_Name"], r["Counter_Value"],
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    for short, cs in agg.items():
        d = summary["kernels"].setdefault(short, {"launch_config": meta[short], "counters_avg_per_dispatch": {}})
        for c, v in cs.items():
            d["counters_avg_per_dispatch"][c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
with open(prefix + "_pmc_dispatches.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["pass", "dispatch", "kernel", "counter", "value", "duration_ns"])
    w.writerows(rows_out)
for k, d in summary["kernels"].items():
    c = {n: v["mean"] for n, v in d["counters_avg_per_dispatch"].items()}
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        d["derived"] = {"wait_any_frac": c.get("SQ_WAIT_ANY", 0) / wc,
                        "wait_inst_any_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                        "active_inst_any_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc}
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
        d.setdefault("derived", {})["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        d.setdefault("derived", {})["hbm_bytes_per_dispatch_raw"] = (c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
with open(prefix + "_pmc_summary.json", "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps(summary, indent=1)[:3000])
