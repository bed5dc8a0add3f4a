#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of `python3 <args>`; prints the per-kernel table.  usage: tools/kstats.sh <tag> <script> [args...]
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/ks_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/$@ > "$OUT/run.log" 2>&1
echo "rc=$?"
cd "$ROOT"
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")[:90]
    print(f'{n:90s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:10.1f} us  {r["Percentage"]}%')
PY
