"""The TD path's two kernels as one profiling workload (rocprofv3 --kernel-trace --stats -- python3 tools/td_profile_workload.py):
K3 at N = 4,096 / 65,536 in the matrix-core order (23 launches each), at N = 4,096 in the reference order (8 launches: kernel
tucker_objective_ref_kernel) and the device Powell on BASELINE config 3 in both objective orders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights

dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "outputs", "features"))
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
for N in (4096, 65536):
    P = torch.from_numpy(synth.tucker_params(N)).to(dev)
    X = torch.from_numpy(synth.features(N, 1404, 3)).to(dev)
    for _ in range(23):
        ops.tucker_objective(Wm, X, P, cp, order="fast")
    torch.cuda.synchronize()
    if N == 4096:
        for _ in range(8):
            ops.tucker_objective(Wm, X, P, cp, order="reference")
        torch.cuda.synchronize()
idx = synth.tucker_grid_indices(4096, seed=2)
Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)
for order in ("fast", "reference"):
    for _ in range(2):
        ops.tucker_powell(Wm, Xg, cp, order=order)
    torch.cuda.synchronize()
print("done")
