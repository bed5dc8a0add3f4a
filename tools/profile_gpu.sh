#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel trace + separate PMC passes
# of the same bench.py command.  Outputs go under gpurun_out/prof_<tag>/; summarise afterwards with
# tools/summarize_prof.py and commit the summaries under profiles/.
set -u
TAG=${1:-r01}
EXTRA_ARGS=${2:-}          # e.g. "--mode f32"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --no-cpu-baseline --no-extra $EXTRA_ARGS"   # default --steps 200 --warmup 50
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- $CMD > "$OUT/pmc_$N.log" 2>&1
  echo "pmc $N rc=$?"
done
cd "$ROOT"
find "$OUT" -name "*.csv" | head -50
du -sh "$OUT"
