#!/usr/bin/env python3
"""gpurun_out/pmcq_<tag>/ (tools/pmc_quick.sh <tag> tools/td_profile_workload.py) -> profiles/<tag>_kernel_stats.csv + profiles/<tag>_summary.md:
durations per (kernel, grid) from the kernel trace (the first three launches of a size dropped) and the mean of every counter per
(kernel, grid), each counter group from its own --pmc pass, with the derived figures DESIGN.md quotes.
usage: python tools/summarize_td_prof.py [tag=r03td] [config-3 evaluations, e.g. 6.674e6]"""
import csv, glob, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03td"
powell_evals = float(sys.argv[2]) if len(sys.argv) > 2 else 6.674e6        # sum of nfev over config 3's 4,096 faces (tools/powell_nfev_dump.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", f"pmcq_{tag}"), os.path.join(ROOT, "profiles")
OPS, ISSUE, F64_MFMA = 135 * 1404 * 5, 39.3e12, 78.6e12


def short(n):
    return n.split("(")[0].replace("void ", "").strip()


for f in glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))

out = [f"# TD path kernels ({tag}: tools/pmc_quick.sh {tag} tools/td_profile_workload.py; this file: tools/summarize_td_prof.py)", "",
       "`rocprofv3 --kernel-trace --stats` and six separate `--pmc` passes of the same workload: K3 at N = 4,096 / 65,536 in the",
       "matrix-core order (23 launches each), K3 at N = 4,096 in the REFERENCE order (8 launches, kernel `tucker_objective_ref_kernel`,",
       "768 threads per workgroup) and the device Powell on BASELINE config 3 (4,096 faces) in both orders (`tucker_powell_kernel<0>` =",
       "matrix-core, `<1>` = reference order, the default).  Durations: kernel trace, the first three launches of a size dropped;",
       "a BURST after idle (the sustained rates are in the `*_sustained_kernel_stats.csv` next to this file).", "",
       "| kernel | grid (threads) | launches | mean us | min us | max us | note |", "|---|---|---|---|---|---|---|"]
mean_us = {}
for key in sorted(dur):
    v = [d for _, d in sorted(dur[key])]
    v = v[3:] if len(v) > 5 else v
    m = sum(v) / len(v) / 1e3
    mean_us[key] = m
    k, g = key
    note = ""
    if k.endswith("tucker_objective_kernel"):
        N = g // 32
        note = f"N = {N}: {100 * N * 2 * 135 * 1404 / (m * 1e-6) / F64_MFMA:.1f} % of the f64 matrix peak"
    elif k.endswith("tucker_objective_ref_kernel"):
        N = g // 768 * 16
        r = N / (m * 1e-6)
        note = f"N = {N}: {r / 1e6:.2f} M evaluations/s = {r * OPS / ISSUE:.3f} of the f64 vector issue rate (39.3 T op/s)"
    elif "tucker_powell_kernel<1>" in k:
        r = powell_evals / (m * 1e-6)
        note = f"config 3, reference order (parity mode, default): {powell_evals / 1e6:.3f} M evaluations -> {r / 1e6:.1f} M evaluations/s = {r * OPS / ISSUE:.3f} of the issue rate"
    elif "tucker_powell_kernel<0>" in k:
        note = "config 3, matrix-core order (opt-in)"
    out.append(f"| `{k}` | {g} | {len(v)} | {m:.1f} | {min(v) / 1e3:.1f} | {max(v) / 1e3:.1f} | {note} |")
out += ["", "## Counters (mean per launch; each group from its own pass)", ""]
for key in sorted(cnt):
    k, g = key
    if "tucker" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in cnt[key].items()}
    out.append(f"### `{k}`, grid {g} ({mean_us.get(key, float('nan')):.1f} us per launch in the trace)")
    for n in sorted(c):
        out.append(f"- {n}: {c[n]:,.1f}")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out.append(f"- **HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE)*1024 = {(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e6:.1f} MB**")
    if "TCC_HIT_sum" in c:
        out.append(f"- **L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}**")
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        out.append(f"- **vector-ALU issue: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8 = {cyc:,.0f} cycles) = {c['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.3f} of the issue slots over the whole launch**")
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
        out.append(f"- **SQ_WAIT_ANY / SQ_WAVE_CYCLES = {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f}; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}**")
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        out.append(f"- **LDS bank-conflict cycles / LDS active = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}**")
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        out.append(f"- **MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024) = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}**")
    out.append("")
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(out))
print("\n".join(out[:24]))
