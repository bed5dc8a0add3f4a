"""Bits of K2 from ONE library build, for comparing builds: runs mode <mode> on a few batch shapes (fused landmarks path normalised and not,
features path with a padded row stride and an odd base alignment) and saves every output to <out>.npz.  Run it once per build
(NLML_HPE_LIB=<path to .so> for an experiment build), then `k2_dump.py --compare a.npz b.npz` prints the arrays that differ (exact).
usage: k2_dump.py <mode 0|1|2|3> <out.npz>   |   k2_dump.py --compare a.npz b.npz"""
import os, sys
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
        bad += not same
        print(f"{k:40s} {'identical' if same else 'DIFFERENT: %d of %d values' % (int((a[k] != b[k]).sum()), a[k].size)}")
    print("ALL IDENTICAL" if bad == 0 else f"{bad} arrays differ")
    sys.exit(1 if bad else 0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nlml_hpe_amd import ops, synth, weights
mode, out = int(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, 0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, mode)).to(dev)
res = {}
for B in (1, 63, 64, 130, 700, 8191, 16384):
    raw = torch.from_numpy(synth.raw_landmarks(B, 7 + B)).to(dev)
    for norm in (True, False):
        pose, lat, valid = ops.landmarks_to_pose(raw, blob, norm, return_latent=True, return_valid=True)
        res[f"fused_B{B}_norm{int(norm)}_pose"] = pose.cpu().numpy()
        res[f"fused_B{B}_norm{int(norm)}_latent"] = lat.cpu().numpy()
        res[f"fused_B{B}_norm{int(norm)}_valid"] = valid.cpu().numpy()
    feats = ops.normalize_ipd(raw, True)
    # a padded row stride (1408) and, through the column offset, every 16-byte phase of the first row
    for off in (0, 4, 12, 28):
        buf = torch.zeros((B, 1408 + 32), dtype=torch.float32, device=dev)
        view = buf[:, off:off + 1404]
        view.copy_(feats)
        res[f"features_B{B}_off{off}_pose"] = ops.encoder_heads_fwd(view, blob, 1404).cpu().numpy()
torch.cuda.synchronize()
np.savez(out, **res)
print("saved", out, len(res), "arrays")
