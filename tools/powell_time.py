"""Time the device Powell on BASELINE config 3 (4,096 grid faces).  usage: python tools/powell_time.py [order]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
dev = torch.device("cuda:0")
order = sys.argv[1] if len(sys.argv) > 1 else "fast"
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
idx = synth.tucker_grid_indices(4096, seed=2)
Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)
ops.tucker_powell(Wm, Xg[:64], cp, order=order)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    res = ops.tucker_powell(Wm, Xg, cp, order=order)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nf = res["nfev"].double()
    print(f"{os.environ.get('NLML_HPE_LIB', 'default')} order={order}: {dt*1e3:.1f} ms  {float(nf.sum())/dt/1e6:.1f} M face-evals/s  mean nfev {float(nf.mean()):.0f} max {int(nf.max())}", flush=True)
