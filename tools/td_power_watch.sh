#!/bin/bash
# usage (GPU box): tools/td_power_watch.sh : K3 in the reference order (N = 4,096) in a ~7 s loop, rocm-smi power / clock sampled three times
python3 - > gpurun_out/pw_td.txt 2>&1 <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
N = 4096
P = torch.from_numpy(synth.tucker_params(N)).to(dev)
X = torch.from_numpy(synth.features(N, 1404, 3)).to(dev)
for _ in range(5): ops.tucker_objective(Wm, X, P, cp, order="reference")
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 7.0:
    for _ in range(200): ops.tucker_objective(Wm, X, P, cp, order="reference")
    torch.cuda.synchronize(); n += 200
dt = time.perf_counter() - t0
print(f"K3 reference order N={N}: {n} launches, {dt / n * 1e6:.1f} us per launch, {N * n / dt / 1e6:.2f} M evaluations/s")
PY
PID=$!
sleep 4.0
for i in 1 2 3; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (edge|junction)" | head -6
  echo ---
  sleep 0.8
done
wait $PID
tail -1 gpurun_out/pw_td.txt
