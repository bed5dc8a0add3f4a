#!/usr/bin/env python3
"""gpurun_out/pmcq_<src>/ (tools/pmc_quick.sh <src> <workload ...>) -> profiles/<tag>_kernel_stats.csv + profiles/<tag>_summary.md.
Per (kernel, grid): launches, mean / min / max duration from the kernel trace (the first `skip` launches of each dropped), the mean
of every counter (each group from its own --pmc pass) and the derived figures: HBM-side bytes per launch = (2*FETCH_SIZE +
WRITE_SIZE)*1024 (MI355X_MICROARCH.md, HBM section), L2 hit rate, MFMA-busy fraction.
usage: python tools/summarize_pmcq.py <src> <tag> [skip=3] ["title"]"""
import csv, glob, os, shutil, sys
from collections import defaultdict

src_tag, tag = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
title = sys.argv[4] if len(sys.argv) > 4 else src_tag
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", f"pmcq_{src_tag}"), os.path.join(ROOT, "profiles")


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
out = [f"# {title} ({tag}; tools/pmc_quick.sh + tools/summarize_pmcq.py)", "",
       "`rocprofv3 --kernel-trace --stats` and separate `--pmc` passes of the same workload.  Durations: kernel trace, the first "
       f"{skip} launches of each (kernel, grid) dropped.", "",
       "| kernel | grid (threads) | launches | mean us | min us | max us |", "|---|---|---|---|---|---|"]
for key in sorted(dur):
    v = [d for _, d in sorted(dur[key])]
    v = v[skip:] if len(v) > skip + 2 else v
    out.append(f"| `{key[0]}` | {key[1]} | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} |")
out += ["", "## Counters (mean per launch; each group from its own pass)", ""]
for key in sorted(cnt):
    m = {c: sum(v) / len(v) for c, v in cnt[key].items()}
    out.append(f"### `{key[0]}`, grid {key[1]}")
    out.append("")
    for c, v in sorted(m.items()):
        out.append(f"- {c}: {v:,.1f}")
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        out.append(f"- **HBM-side bytes per launch (corrected)**: {(2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024:,.0f}")
    if "TCC_HIT_sum" in m and m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0) > 0:
        out.append(f"- **L2 hit rate**: {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        out.append(f"- **MFMA-busy fraction** (busy cycles / (GUI-active cycles / 8 XCDs x 1,024 SIMDs)): {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
    if "SQ_WAIT_ANY" in m and m.get("SQ_BUSY_CYCLES"):
        pass
    if "SQ_WAIT_INST_ANY" in m and m.get("SQ_WAVE_CYCLES"):
        out.append(f"- **SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES**: {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"- **LDS bank-conflict cycles / LDS-active cycles**: {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
    out.append("")
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
