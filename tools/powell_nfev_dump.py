"""Dump the per-face evaluation counts of the device Powell on BASELINE config 3 (for the round-cost model in DESIGN.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
idx = synth.tucker_grid_indices(4096, seed=2)
Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)
order = sys.argv[1] if len(sys.argv) > 1 else "reference"
res = ops.tucker_powell(Wm, Xg, cp, order=order)
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/powell_nfev_{order}.npy", res["nfev"].cpu().numpy())
print("saved", res["nfev"].shape)
