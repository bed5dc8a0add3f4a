"""Development aid: split-f16 mode vs the f64 oracle + timing (imports the oracle, hence under tests/; never part of the product path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights, _lib
from oracle import encoder_heads as eh, feature_norm as fn

dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts("models")
for F, B in ((1404, 1000), (136, 333), (13, 70)):
    sd = synth.encoder_state_dict(F, 0)
    P = eh.Params(sd, heads)
    blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2)).to(dev)
    x = synth.features(B, F, 1)
    out = ops.encoder_heads_fwd(torch.from_numpy(x).to(dev), blob, F)
    torch.cuda.synchronize()
    ref = eh.forward_numpy(x, P, np.float64)
    e = np.abs(np.degrees(out.cpu().numpy().astype(np.float64) - ref))
    print(f"features F={F} B={B}: max {e.max():.3e} deg mean {e.mean():.3e}  nan={np.isnan(e).sum()}", flush=True)
sd = synth.encoder_state_dict(1404, 0)
P = eh.Params(sd, heads)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2)).to(dev)
blob32 = torch.from_numpy(weights.pack_blob(sd, heads)).to(dev)
raw = synth.raw_landmarks(777, 1)
out = ops.landmarks_to_pose(torch.from_numpy(raw).to(dev), blob, True)
out = out[0] if isinstance(out, tuple) else out
ref = eh.forward_numpy(fn.normalize_ipd(raw, True), P, np.float64)
e = np.abs(np.degrees(out.cpu().numpy().astype(np.float64) - ref))
print(f"fused B=777: max {e.max():.3e} deg mean {e.mean():.3e}", flush=True)
B = 65536
raw = torch.from_numpy(synth.raw_landmarks(B, 1)).to(dev)
feats = ops.normalize_ipd(raw, True)
for name, bl in (("f16x2", blob), ("f32", blob32)):
    for label, fn_ in (("fused", lambda: ops.landmarks_to_pose(raw, bl, True)), ("features", lambda: ops.encoder_heads_fwd(feats, bl, 1404))):
        for _ in range(3): fn_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn_()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name} {label}: {ms:.3f} ms  {B/ms*1e3/1e6:.2f} Mfaces/s", flush=True)
a = ops.landmarks_to_pose(raw, blob, True); b = ops.landmarks_to_pose(raw, blob32, True)
a = a[0] if isinstance(a, tuple) else a; b = b[0] if isinstance(b, tuple) else b
d = torch.rad2deg((a - b).abs())
print(f"f16x2 vs f32 kernel on 65536 faces: max {d.max().item():.3e} deg mean {d.mean().item():.3e}")
