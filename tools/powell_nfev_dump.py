"""Dump per-face nfev of the device Powell run on the BASELINE config-3 faces (analysis aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts(os.path.join(ROOT, "outputs", "features"))
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
idx = synth.tucker_grid_indices(4096, seed=2)
Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)
ops.tucker_powell(Wm, Xg[:64], cp); torch.cuda.synchronize()
t0 = time.perf_counter(); res = ops.tucker_powell(Wm, Xg, cp); torch.cuda.synchronize(); dt = time.perf_counter() - t0
nf = res["nfev"].cpu().numpy()
np.save(os.path.join(ROOT, "gpurun_out", "powell_nfev.npy"), nf)
print(f"{dt:.4f} s, mean nfev {nf.mean():.0f}, max {nf.max()}")
