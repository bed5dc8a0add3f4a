"""Development aid: layer-per-launch small-batch path vs the fused split-f16 kernel (bit equality) and timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights, _lib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(R, "models"))
for F, B in ((1404, 64), (1404, 200), (1404, 2000), (136, 77), (13, 5)):
    sd = synth.encoder_state_dict(F, 0)
    blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2)).to(dev)
    x = torch.from_numpy(synth.features(B, F, 1)).to(dev)
    a, la, va = ops.encoder_heads_fwd(x, blob, F, return_latent=True, return_valid=True)
    b, lb, vb = ops.encoder_heads_fwd_small(x, blob, F, return_latent=True, return_valid=True)
    torch.cuda.synchronize()
    print(f"features F={F} B={B}: pose equal {torch.equal(a, b)} (max diff {(a-b).abs().max().item():.3e}), latent equal {torch.equal(la, lb)}, valid equal {torch.equal(va, vb)}", flush=True)
sd = synth.encoder_state_dict(1404, 0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2)).to(dev)
for B in (64, 131, 2000, 4096, 16384):
    raw = torch.from_numpy(synth.raw_landmarks(B, 3)).to(dev)
    a = ops.landmarks_to_pose(raw, blob, True); b = ops.landmarks_to_pose_small(raw, blob, True)
    eq = torch.equal(a, b)
    res = []
    for fn in (lambda: ops.landmarks_to_pose(raw, blob, True), lambda: ops.landmarks_to_pose_small(raw, blob, True)):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 100)
    print(f"fused-landmarks B={B}: equal {eq}  fused {res[0]*1e3:.1f} us  layered {res[1]*1e3:.1f} us", flush=True)
