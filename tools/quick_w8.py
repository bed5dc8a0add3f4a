"""First contact of the eight-wave strict kernel with the hardware: one small launch (wrapped in `timeout` by the caller), then bits
against the layer-per-launch path and accuracy against the f64 oracle on a few shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nlml_hpe_amd import _lib, ops, synth, weights
from oracle import encoder_heads as EH
from oracle import feature_norm as FN
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2S)).to(dev)
bad = 0
for B in (64, 1, 100, 4096 + 37):
    raw_np = synth.raw_landmarks(B, seed=5)
    raw = torch.from_numpy(raw_np).to(dev)
    out, lat, val = ops.landmarks_to_pose(raw, blob, True, return_latent=True, return_valid=True)
    torch.cuda.synchronize()
    print("launched", B, flush=True)
    o2, l2, v2 = ops.landmarks_to_pose_small(raw, blob, True, return_latent=True, return_valid=True)
    ref = EH.forward_numpy(FN.normalize_ipd(raw_np, True), EH.Params(sd, heads), np.float64)
    err = np.degrees(np.abs(out.cpu().numpy() - ref)).max()
    print(f"B={B}: fused == layered {torch.equal(out, o2)} latent {torch.equal(lat, l2)} valid {torch.equal(val, v2)}; max err vs f64 {err:.2e} deg", flush=True)
    feats = ops.normalize_ipd(raw, True)
    o3 = ops.encoder_heads_fwd(feats, blob, 1404)
    print("   features path == fused:", torch.equal(o3, out), flush=True)
    ok = torch.equal(out, o2) and torch.equal(lat, l2) and torch.equal(val, v2) and torch.equal(o3, out) and err < 1e-4
    bad = bad + (0 if ok else 1)
print("QUICK_W8", "PASS" if bad == 0 else f"FAIL ({bad} shapes)")
