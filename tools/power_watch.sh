#!/bin/bash
# usage (GPU box): tools/power_watch.sh <mode 0|1|2> <fused|features> : run K2 in a loop ~6 s and sample rocm-smi power/clock
python3 tools/k2_mode_run.py $1 $2 65536 4000 > gpurun_out/pw_run_$1.txt 2>&1 &
PID=$!
sleep 2.5
for i in 1 2 3; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (edge|junction)" | head -6
  echo ---
  sleep 0.8
done
wait $PID
cat gpurun_out/pw_run_$1.txt | tail -1
