"""K3 at its SUSTAINED rate: N = 4,096 (one workgroup per CU), both operation orders, 400 untimed launches (the clock needs tens of
milliseconds of load to reach its sustained level: short bursts after idle run at ~2.0 GHz, the sustained loop at the board's power
limit) and then 1,000 launches back to back between one event pair.  Also the profiling workload of profiles/r03td_sustained_*:
    rocprofv3 --kernel-trace --stats -- python3 tools/td_sustained.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights

dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
art = weights.load_tucker_artefacts(os.path.join(ROOT, "outputs", "features"))
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
N = 4096
P = torch.from_numpy(synth.tucker_params(N)).to(dev)
X = torch.from_numpy(synth.features(N, 1404, 3)).to(dev)
for order, per_eval, peak, unit in (("reference", 135 * 1404 * 5, 39.3e12, "of the f64 vector issue rate (39.3 T op/s)"),
                                    ("fast", 383.7e3, 78.6e12, "of the f64 matrix peak (78.6 TFLOP/s)")):
    for _ in range(400): ops.tucker_objective(Wm, X, P, cp, order=order)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): ops.tucker_objective(Wm, X, P, cp, order=order)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)
    r = N / (us * 1e-6)
    print(f"K3 {order} order, N={N}, sustained: {us:.1f} us per launch  {r / 1e6:.2f} M evaluations/s  {r * per_eval / peak:.3f} {unit}", flush=True)

# the device Powell on BASELINE config 3 straight after that load (clock already at its sustained level) and after an idle second
import time
idx = synth.tucker_grid_indices(4096, seed=2)
Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)
for label, pre in (("after 400 K3 launches", lambda: [ops.tucker_objective(Wm, X, P, cp, order="reference") for _ in range(400)]),
                   ("after 1 s of idle", lambda: time.sleep(1.0))):
    for order in ("reference", "fast"):
        pre(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ops.tucker_powell(Wm, Xg, cp, order=order)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"Powell {order} order, 4096 faces, {label}: {dt * 1e3:.1f} ms  {float(res['nfev'].double().sum()) / dt / 1e6:.1f} M face-evaluations/s", flush=True)
