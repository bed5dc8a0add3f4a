"""Per-round cost of the device Powell: one face alone (vector-ALU rounds) and 16 copies of it (matrix-core rounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
dev = torch.device("cuda:0")
order = sys.argv[1] if len(sys.argv) > 1 else "fast"
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
idx = synth.tucker_grid_indices(64, seed=2)
Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)
for copies in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16):
    X = Xg[:1].repeat(copies, 1).contiguous()
    ops.tucker_powell(Wm, X, cp, order=order); torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ops.tucker_powell(Wm, X, cp, order=order)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nf = int(res["nfev"].max())
    print(f"order={order} live machines {copies:2d}: {dt*1e3:7.2f} ms for {nf} rounds = {dt/nf*1e6:6.2f} us per round", flush=True)
