"""Where a 128-face tile's cycles go in the wide path's layer kernels (-DWIDE_STAMPS build, NLML_HPE_LIB pointing at it): per pass
prologue / K loop / epilogue, per wave class (waves 0-3: the older wave of each SIMD, 4-7: the younger).
usage: NLML_HPE_LIB=exp_libs/wstamps.so python tools/wide_stamps.py [stage 0|1|2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
stamps = torch.zeros((B // 128, 8, 32), dtype=torch.int64, device=dev)
os.environ["NLML_WIDE_STAMPS_PTR"] = hex(stamps.data_ptr())
os.environ["NLML_WIDE_STAMPS_STAGE"] = str(stage)
from nlml_hpe_amd import _lib, synth, weights
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2S)).to(dev)
L = _lib.lib()
raw = torch.from_numpy(synth.raw_landmarks(B, seed=1)).to(dev)
ws = torch.empty((L.nlml_encoder_heads_workspace_bytes(B, 1404),), dtype=torch.uint8, device=dev)
out = torch.empty((B, 3), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(60):
    L.nlml_landmarks_to_pose_wide(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64)
npass = {0: 4, 1: 2, 2: 1}[stage]
print(f"stage {stage}: {npass} passes; s_memtime ticks (100 MHz-constant clock on gfx950? see below), mean over {s.shape[0]} tiles")
t0 = s[:, :, 0].min(axis=1, keepdims=True)
print("tile span (first stamp of any wave -> last stamp):", (s[:, :, 8 * (npass - 1) + 3].max(axis=1) - t0[:, 0]).mean())
for p in range(npass):
    b = 8 * p
    for name, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        pro = (s[:, sl, b + 1] - s[:, sl, b + 0]).mean()
        loop = (s[:, sl, b + 2] - s[:, sl, b + 1]).mean()
        epi = (s[:, sl, b + 3] - s[:, sl, b + 2]).mean()
        print(f"  pass {p} {name}: prologue {pro:9.0f}  K loop {loop:9.0f}  epilogue {epi:9.0f}")
# first-round tiles (blocks 0..255) against second-round ones
for name, sl in (("blocks 0-255", slice(0, 256)), ("blocks 256-511", slice(256, 512)))[:0]:
    print(name, "K loop of pass 0, waves 4-7:", (s[sl, 4:, 2] - s[sl, 4:, 1]).mean(), " start offset vs block 0:", (s[sl, 0, 0] - s[0, 0, 0]).mean())
