#!/bin/bash
# usage (on the GPU box): tools/pmc_quick.sh <tag> <python script + args...> : kernel trace + two PMC passes
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/$@ > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 180 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 $ROOT/$@ > "$OUT/pmc_$N.log" 2>&1
  echo "pmc $N rc=$?"
done
cd "$ROOT"
python3 - <<PY
import csv, glob, collections
c = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in c.items():
    if not any(t in k for t in ("encoder_heads", "tucker", "wide_layer", "tail64", "tail_ws")): continue
    print(k)
    for n, v in sorted(cs.items()): print(f"   {n:32s} {sum(v)/len(v):18,.1f}  (n={len(v)})")
for f in glob.glob("$OUT/trace/**/*_kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
