import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from nlml_hpe_amd import _lib, ops, synth, weights
from test_gpu_parity import _wide_call
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2S)).to(dev)
for B in (700, 129, 257, 640, 641, 1000):
    raw_np = synth.raw_landmarks(B, seed=5)
    raw_np[7] = 0.0
    raw_np[B - 1] = raw_np[3]
    raw = torch.from_numpy(raw_np).to(dev)
    for normalize in (True, False):
        for rep in range(3):
            o, l, v = ops.landmarks_to_pose(raw, blob, normalize, return_latent=True, return_valid=True)
            o2, l2, v2 = _wide_call("nlml_landmarks_to_pose_wide", raw, B, blob, dev, extra=(B, int(normalize)))
            do = (o != o2).any(dim=1).nonzero().flatten().tolist()
            dl = (l != l2).any(dim=1).nonzero().flatten().tolist()
            dv = (v.to(torch.uint8) != v2).nonzero().flatten().tolist()
            print(f"B={B} norm={normalize} rep={rep}: pose rows differ {do[:10]} ({len(do)}), latent {dl[:10]} ({len(dl)}), valid {dv[:10]} ({len(dv)})", flush=True)
            if do:
                i = do[0]
                print("   fused", o[i].tolist(), "wide", o2[i].tolist(), "raw row absmax", float(raw[i].abs().max()))
