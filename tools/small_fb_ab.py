"""The layer-per-launch path with one unit per face block for the smallest batches against units of both face blocks
(NLML_K2_SMALL_FB2=1, read once per process: run this script twice): latency at 1 .. 1,024 faces in both split modes, bits against the
fused kernel.   python tools/small_fb_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nlml_hpe_amd import _lib, ops, synth, weights
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
ok = True
for mode in (_lib.MODE_F16X2S, _lib.MODE_F16X2):
    blob = torch.from_numpy(weights.pack_blob(sd, heads, mode)).to(dev)
    for B in [int(x) for x in os.environ.get('FB_SIZES', '1,64,128,256,512,1024').split(',')]:
        raw = torch.from_numpy(synth.raw_landmarks(B, seed=7)).to(dev)
        a, la, va = ops.landmarks_to_pose(raw, blob, True, return_latent=True, return_valid=True)
        b, lb, vb = ops.landmarks_to_pose_small(raw, blob, True, return_latent=True, return_valid=True)
        same = torch.equal(a, b) and torch.equal(la, lb) and torch.equal(va, vb)
        ok &= same
        for _ in range(50):
            ops.landmarks_to_pose_small(raw, blob, True)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for e0, e1 in evs:
            e0.record(); ops.landmarks_to_pose_small(raw, blob, True); e1.record()
        torch.cuda.synchronize()
        ms = float(np.median([e0.elapsed_time(e1) for e0, e1 in evs]))
        print(f"fb2={os.environ.get('NLML_K2_SMALL_FB2', '0')} mode={mode} B={B}: {ms * 1e3:.1f} us per call, layered == fused: {same}", flush=True)
print("SMALL_FB", "PASS" if ok else "FAIL")
