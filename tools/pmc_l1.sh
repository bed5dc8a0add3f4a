#!/bin/bash
# usage (on the GPU box): tools/pmc_l1.sh <tag> <python script + args...>
# The vector-L1 (TCP) / texture-path / L2 (TCC) counters of a workload, one `rocprofv3 --pmc` pass per group (never together with a trace),
# preceded by `rocprofv3 --list-avail` so that a name this build of the profiler does not know shows up in the log instead of as an empty pass.
# Output: gpurun_out/pmcl1_<tag>/ + a table on stdout (mean per launch by kernel).
# (No TA_* / TD_* group: with those rocprofv3 aborted after the workload and sat until its timeout, twice -- round 5.)
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmcl1_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 120 rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
echo "list rc=$?"
GROUPS_=(
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum"
  "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum"
  "TCC_REQ_sum TCC_READ_sum TCC_BUSY_sum TCC_TAG_STALL_sum"
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
  "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
  "SQ_INSTS_LDS SQ_INST_CYCLES_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"
)
i=0
for C in "${GROUPS_[@]}"; do
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$i" -- python3 $ROOT/$@ > "$OUT/pmc_$i.log" 2>&1
  echo "pmc group $i ($C) rc=$?"
  i=$((i + 1))
done
cd "$ROOT"
python3 - <<PY
import csv, glob, collections
c = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in c.items():
    if not any(t in k for t in ("encoder_heads", "tucker", "wide_layer", "tail64", "layer_kernel", "tail_ws")): continue
    print(k)
    for n, v in sorted(cs.items()): print(f"   {n:40s} {sum(v)/len(v):20,.1f}  (n={len(v)})")
PY
