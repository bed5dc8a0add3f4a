#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by tools/profile_gpu.sh) into profiles/<tag>_*.

  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats table, verbatim
  profiles/<tag>_pmc_summary.md/json  per-kernel mean of every collected counter + derived figures
  profiles/<tag>_pmc_traffic.json     HBM-side bytes per launch of the dominant kernel, corrected as
                                      MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE are in
                                      KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream
                                      => bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024   (read by bench.py)
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
path_key = sys.argv[2] if len(sys.argv) > 2 else "f16x2_fused"      # <mode>_<path> of the profiled bench.py run
traffic_tag = sys.argv[3] if len(sys.argv) > 3 else tag             # bench.py reads profiles/<latest round>_pmc_traffic.json
bench_args = sys.argv[4] if len(sys.argv) > 4 else ""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


counters = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        counters[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
durs = defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "**", "*_kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        durs[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))

summary = {}
for k, cs in counters.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"launches_profiled": max(len(v) for v in cs.values()), "counters_mean": m}
    if k in durs:
        d["avg_duration_ns_trace"] = sum(durs[k]) / len(durs[k])
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        d["hbm_bytes_per_launch_corrected"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in m:
        d["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # MFMA-busy cycles summed over all SIMDs vs (kernel-active cycles, summed over the 8 XCDs) * 128 SIMDs per XCD
        d["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m:
        d["wait_any_frac_of_wave_cycles"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_INST_ANY" in m:
        d["wait_inst_frac_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    summary[k] = d
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)

with open(os.path.join(dst, f"{tag}_pmc_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary {tag}\n\nCommand: `python3 bench.py --no-cpu-baseline --no-extra {bench_args}` (default 50 warm-up + 200 timed steps) "
            "(kernel trace and each --pmc group in separate passes; tools/profile_gpu.sh).\n\n")
    for k, d in summary.items():
        f.write(f"## {k}\n\n")
        for kk, vv in d.items():
            if kk == "counters_mean":
                for c, v in sorted(vv.items()):
                    f.write(f"- {c}: {v:,.1f}\n")
            else:
                f.write(f"- **{kk}**: {vv:,.6g}\n")
        f.write("\n")

dom = max(summary, key=lambda k: summary[k].get("avg_duration_ns_trace", 0) * summary[k]["launches_profiled"]) if summary else None
if dom and "hbm_bytes_per_launch_corrected" in summary[dom]:
    tp = os.path.join(dst, f"{traffic_tag}_pmc_traffic.json")
    cur = json.load(open(tp)) if os.path.exists(tp) else {}
    cur[f"{path_key}_bytes_per_launch"] = summary[dom]["hbm_bytes_per_launch_corrected"]
    cur[f"{path_key}_kernel"] = dom
    if "avg_duration_ns_trace" in summary[dom]:
        cur[f"{path_key}_kernel_ms"] = summary[dom]["avg_duration_ns_trace"] / 1e6
    cur["correction"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024, MI355X_MICROARCH.md HBM section"
    json.dump(cur, open(tp, "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc_summary.md")).read())
