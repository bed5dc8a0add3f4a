"""First contact of the trunk + streamed-tail path (encoder_heads_f16x2_w8.hip TRUNK, encoder_heads_f16x2_tailws.hip) with the hardware: bits
against the fused eight-wave kernel on a few batch sizes (raw landmarks and features), then timings of both at 65,536 faces.  Run under
`timeout` (caller)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nlml_hpe_amd import _lib, ops, synth, weights

dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2S)).to(dev)
L = _lib.lib()


def st_raw(raw, normalize=True, want=True):
    B = raw.shape[0]
    ws = torch.empty((L.nlml_encoder_heads_workspace_bytes(B, 1404),), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 3), dtype=torch.float32, device=dev)
    lat = torch.empty((B, 9), dtype=torch.float32, device=dev) if want else None
    val = torch.empty((B,), dtype=torch.uint8, device=dev) if want else None
    _lib.check(L.nlml_landmarks_to_pose_streamed(raw.data_ptr(), B, int(normalize), blob.data_ptr(), blob.numel(), out.data_ptr(),
                                             lat.data_ptr() if want else None, val.data_ptr() if want else None,
                                             ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "streamed")
    return out, lat, val


def st_feats(x):
    B = x.shape[0]
    ws = torch.empty((L.nlml_encoder_heads_workspace_bytes(B, 1404),), dtype=torch.uint8, device=dev)
    out = torch.empty((B, 3), dtype=torch.float32, device=dev)
    _lib.check(L.nlml_encoder_heads_fwd_streamed(x.data_ptr(), x.stride(0), B, 1404, blob.data_ptr(), blob.numel(), out.data_ptr(),
                                             None, None, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "streamed")
    return out


bad = 0
for B in (128, 1, 200, 4096 + 37, 16384, 65536 - 63):
    raw_np = synth.raw_landmarks(B, seed=5)
    if B >= 200:
        raw_np[7] = 0.0          # a "no face" row
    raw = torch.from_numpy(raw_np).to(dev)
    o2, l2, v2 = st_raw(raw)
    torch.cuda.synchronize()
    print("launched", B, flush=True)
    out, lat, val = ops.landmarks_to_pose(raw, blob, True, return_latent=True, return_valid=True)
    eq = (torch.equal(out, o2), torch.equal(lat, l2), torch.equal(val.to(torch.uint8), v2))
    feats = ops.normalize_ipd(raw, True)
    o3 = st_feats(feats)
    o4, _, _ = st_raw(raw, normalize=False)
    o5 = ops.landmarks_to_pose(raw, blob, False)
    d = (out - o2).abs().max().item()
    print(f"B={B}: streamed == fused pose {eq[0]} latent {eq[1]} valid {eq[2]} (max |diff| {d:.3e}); features path {torch.equal(o3, out)}; "
          f"un-normalised {torch.equal(o4, o5)}", flush=True)
    bad += 0 if (all(eq) and torch.equal(o3, out) and torch.equal(o4, o5)) else 1
print("QUICK_STREAMED", "PASS" if bad == 0 else f"FAIL ({bad} shapes)", flush=True)

B = 65536
raw = torch.from_numpy(synth.raw_landmarks(B, seed=1)).to(dev)
ws = torch.empty((L.nlml_encoder_heads_workspace_bytes(B, 1404),), dtype=torch.uint8, device=dev)
out = torch.empty((B, 3), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run_streamed():
    L.nlml_landmarks_to_pose_streamed(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st)


def run_fused():
    L.nlml_landmarks_to_pose(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, st)


for name, fn in (("fused", run_fused), ("streamed", run_streamed), ("fused", run_fused), ("streamed", run_streamed)):
    for _ in range(150):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 100
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: {ms:.4f} ms per 65,536 faces = {B / ms / 1e3:.1f} M faces/s", flush=True)
