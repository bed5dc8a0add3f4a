"""Phase breakdown of K3 from the -DK3_STAMPS diagnostic build.  usage: NLML_HPE_LIB=exp_libs/k3_stamps.so python tools/k3_stamps.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = torch.from_numpy(synth.tucker_params(N)).to(dev)
X = torch.from_numpy(synth.features(N, 1404, 3)).to(dev)
for _ in range(3):
    err, xh = ops.tucker_objective(Wm, X, P, cp, return_xhat=True)
torch.cuda.synchronize()
st = xh.cpu().numpy().reshape(-1).view(np.uint64)[: (N // 16) * 8 * 8].reshape(N // 16, 8, 8).astype(np.int64)
d = np.diff(st[:, :, :6], axis=2)
for n, v in zip(["coef (cos, c[q][e], 2 barriers)", "mfma (34 K steps)", "x loads", "residual + barrier", "err store"], d.mean(axis=(0, 1))):
    print(f"{n:34s} {v:9,.0f} cycles")
tot = st[:, :, 5] - st[:, :, 0]
print(f"workgroup total {tot.mean():,.0f} cycles; kernel span {(st[:, :, 5].max() - st[:, :, 0].min()):,} cycles (s_memtime ticks)")
