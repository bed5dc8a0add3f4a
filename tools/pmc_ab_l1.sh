#!/bin/bash
# usage: pmc_two.sh <tag> : TCP request / latency / pending-stall counters for the in-tree build and for exp_libs/xl.so
set -u
ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
for V in intree xl; do
  if [ $V = xl ]; then export NLML_HPE_LIB=$ROOT/exp_libs/xl.so; fi
  for G in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $G | cut -c1-20)
    timeout -k 10 200 rocprofv3 --pmc $G --output-format csv -d $ROOT/gpurun_out/pmc2_$1/${V}_$N -- python3 $ROOT/tools/wide_workload.py fused 30 65536 f16x2s > $ROOT/gpurun_out/pmc2_$1/${V}_$N.log 2>&1
    echo "$V $N rc=$?"
  done
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
for V in ("intree", "xl"):
    c = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc2_$1/%s_*/**/*_counter_collection.csv" % V, recursive=True):
        for r in csv.DictReader(open(f)):
            if "w8_kernel" in r["Kernel_Name"]: c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(V, {k: round(sum(v)/len(v)) for k, v in sorted(c.items())})
PY
