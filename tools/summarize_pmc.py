#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/regen_profiles.sh into what is kept under
profiles/: the kernel-stats table as is, per-dispatch counter rows of OUR kernels only
(the raw collection also lists every rocclr copy kernel), a summary with derived
fractions, and -- when the FETCH_SIZE / WRITE_SIZE passes are there -- the HBM traffic
per dispatch of every kernel, corrected as MI355X_MICROARCH.md prescribes (counters in
KiB; FETCH_SIZE may report half the bytes of wide coalesced reads on gfx950: both the
raw and the doubled figure are kept, the doubled one is the upper bound).

usage: tools/summarize_pmc.py gpurun_out/<dir> profiles/<prefix> [--commit HASH]
                              [--command "..."] [kernel-substring ...]
"""
import collections
import csv
import json
import os
import shutil
import sys

args = sys.argv[1:]
commit, command = "unknown", ""
if "--commit" in args:
    i = args.index("--commit")
    commit = args[i + 1]
    del args[i:i + 2]
if "--command" in args:
    i = args.index("--command")
    command = args[i + 1]
    del args[i:i + 2]
src, prefix = args[0], args[1]
keys = []
for a in args[2:]:
    keys += a.split()
keys = keys or ["ssvs_", "kalman", "atb_mfma", "xtx_mfma"]


def short_name(kn):
    # ("(anonymous namespace)::" goes first: its bracket is not the argument list's)
    return (kn.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            .replace("boom_amd::", ""))


stats = os.path.join(src, "stats", "stats_kernel_stats.csv")
if not os.path.exists(stats):   # (rocprofv3 nests the output under the host name)
    for root, _, files in os.walk(os.path.join(src, "stats")):
        for fn in files:
            if fn.endswith("kernel_stats.csv"):
                stats = os.path.join(root, fn)
with open(stats) as fh:
    body = fh.read()
with open(prefix + "_kernel_stats.csv", "w") as fh:
    fh.write("# commit %s | rocprofv3 --kernel-trace --stats -- python3 %s\n" % (commit, command))
    fh.write(body)
# every launch of the first kernel family of the list, in start order, from the stats run's
# kernel trace: launches that overlap (consecutive sweep launches hand chains over, DESIGN
# section 1) last longer than a lone one, so the stats table's average mixes two things;
# this file keeps them apart
trace = os.path.join(os.path.dirname(stats), os.path.basename(stats).replace("kernel_stats", "kernel_trace"))
if os.path.exists(trace):
    with open(trace) as fh:
        tr = [r for r in csv.DictReader(fh) if keys[0] in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    launches, prev_end = [], 0
    for i, r in enumerate(tr):
        s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nxt = int(tr[i + 1]["Start_Timestamp"]) if i + 1 < len(tr) else None
        launches.append({"kernel": short_name(r["Kernel_Name"]), "ms": round((e0 - s0) / 1e6, 4),
                         "stream": r.get("Stream_Id", ""),
                         "overlaps_another": bool(s0 < prev_end or (nxt is not None and nxt < e0))})
        prev_end = max(prev_end, e0)
    if launches and len(launches) <= 4000:
        per = {}
        for l in launches:
            d = per.setdefault(l["kernel"], {"overlapping": [], "alone": []})
            d["overlapping" if l["overlaps_another"] else "alone"].append(l["ms"])
        means = {k: {"mean_ms_overlapping": round(sum(v["overlapping"]) / len(v["overlapping"]), 4) if v["overlapping"] else None,
                     "n_overlapping": len(v["overlapping"]),
                     "mean_ms_alone": round(sum(v["alone"]) / len(v["alone"]), 4) if v["alone"] else None,
                     "n_alone": len(v["alone"])} for k, v in per.items()}
        with open(prefix + "_launches.json", "w") as fh:
            json.dump({"commit": commit, "command": "python3 " + command, "kernel_family": keys[0],
                       "per_kernel": means, "launches_in_start_order": launches}, fh, indent=1)
summary = {"commit": commit, "command": "python3 " + command, "source": src, "kernels": {}}
rows_out = []
for name in sorted(os.listdir(src)):
    if not name.startswith("pmc_") or not os.path.isdir(os.path.join(src, name)):
        continue
    path = None
    for root, _, files in os.walk(os.path.join(src, name)):
        for fn in files:
            if fn.endswith("counter_collection.csv"):
                path = os.path.join(root, fn)
    if path is None:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"]
        if not any(k in kn for k in keys):
            continue
        short = short_name(kn)
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[short] = dict(grid=r["Grid_Size"], workgroup=r["Workgroup_Size"], lds=r["LDS_Block_Size"],
                           scratch=r["Scratch_Size"], vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"])
        rows_out.append([name, r["Dispatch_Id"], short, r["Counter_Name"], r["Counter_Value"],
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    for short, cs in agg.items():
        d = summary["kernels"].setdefault(short, {"launch_config": meta[short], "counters_avg_per_dispatch": {}})
        for c, v in cs.items():
            d["counters_avg_per_dispatch"][c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
if rows_out:
    # (per-dispatch rows of long runs are many: keep at most 400 per kernel and counter)
    seen = collections.Counter()
    with open(prefix + "_pmc_dispatches.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["# commit " + commit])
        w.writerow(["pass", "dispatch", "kernel", "counter", "value", "duration_ns"])
        for row in rows_out:
            seen[(row[2], row[3])] += 1
            if seen[(row[2], row[3])] <= 400:
                w.writerow(row)
traffic = {}
for k, d in summary["kernels"].items():
    c = {n: v["mean"] for n, v in d["counters_avg_per_dispatch"].items()}
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        d["derived"] = {"wait_any_frac": c.get("SQ_WAIT_ANY", 0) / wc,
                        "wait_inst_any_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                        "active_inst_any_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc}
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
        d.setdefault("derived", {})["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        raw = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        hi = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        d.setdefault("derived", {}).update(hbm_bytes_per_dispatch_raw=raw,
                                           hbm_bytes_per_dispatch_fetch_doubled=hi)
        traffic[k] = {"FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                      "traffic_bytes_raw": raw, "traffic_bytes": hi,
                      "dispatches": d["counters_avg_per_dispatch"]["FETCH_SIZE"]["dispatches"],
                      "scratch_bytes_per_lane": d["launch_config"]["scratch"]}
if summary["kernels"]:
    with open(prefix + "_pmc_summary.json", "w") as fh:
        json.dump(summary, fh, indent=1)
if traffic:
    with open(prefix + "_pmc_traffic.json", "w") as fh:
        json.dump({"commit": commit, "command": "python3 " + command,
                   "method": "separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, per-dispatch "
                             "means; KiB -> bytes; traffic_bytes doubles FETCH_SIZE (MI355X_MICROARCH.md: on "
                             "gfx950 it reports half the bytes of wide coalesced reads; uncalibrated for "
                             "8-byte gathers, so the doubled figure is an upper bound), traffic_bytes_raw "
                             "does not",
                   "kernels": traffic}, fh, indent=1)
print(json.dumps(summary, indent=1)[:2000])
