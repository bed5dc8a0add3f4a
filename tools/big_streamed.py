"""The trunk + streamed-tail path against the fused kernel at batch sizes beyond one round of tail workgroups: 300,001 and 1,048,653 faces
(a "no face" row and an f16-overflow row thrown in): pose, latent, mask bit for bit.  Round 5: [True, True, True] at both sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from nlml_hpe_amd import _lib, ops, synth, weights
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models"))
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(1404, seed=0), heads, _lib.MODE_F16X2S)).to(dev)
for B in (300001, 1048576 + 77):
    base = synth.raw_landmarks(65536, seed=3)
    reps = (B + 65535) // 65536
    raw_np = np.concatenate([base * (1.0 + 0.001 * r) for r in range(reps)], axis=0)[:B].astype(np.float32)
    raw_np[12345] = 0.0
    raw_np[B - 5] *= 3.0e5
    raw = torch.from_numpy(raw_np).to(dev)
    a = ops.landmarks_to_pose(raw, blob, True, return_latent=True, return_valid=True)
    b = ops.landmarks_to_pose_streamed(raw, blob, True, return_latent=True, return_valid=True)
    print(B, [bool(torch.equal(x.view(torch.uint8) if x.dtype != torch.bool else x, y.view(torch.uint8) if y.dtype != torch.bool else y)) for x, y in zip(a, b)], bool(torch.isfinite(b[0]).all()), flush=True)
    del raw, a, b
    torch.cuda.empty_cache()
