"""Where the streamed tail's time goes (-DTW_STAMPS build of encoder_heads_f16x2_tailws.hip: s_memtime stamps per wave and unit in the buffer
passed as `latent`).  usage: NLML_HPE_LIB=<stamp build> python tools/tw_stamps.py   (timing only: no latent in that build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nlml_hpe_amd import _lib, synth, weights

dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2S)).to(dev)
L = _lib.lib()
B = 65536
raw = torch.from_numpy(synth.raw_landmarks(B, seed=1)).to(dev)
ws = torch.empty((L.nlml_encoder_heads_workspace_bytes(B, 1404),), dtype=torch.uint8, device=dev)
out = torch.empty((B, 3), dtype=torch.float32, device=dev)
stamps = torch.zeros((256 * 8 * 96,), dtype=torch.int64, device=dev)
assert stamps.numel() * 8 >= 0
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    _lib.check(L.nlml_landmarks_to_pose_streamed(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), stamps.data_ptr(), None,
                                                 ws.data_ptr(), ws.numel(), st), "streamed")
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(256, 8, 96).astype(np.int64)
names = ["E3.01", "E3.23", "E4", "E5"] + [f"H{k}.{g}" for g in range(3) for k in ("0", "1a", "1b", "1c", "1d", "2ab", "2cd", "3", "4")]
total = (s[:, :, 3 + 2 * (len(names) - 1)] - s[:, :, 0]).mean()
print(f"kernel entry -> last barrier: {total:,.0f} cycles (mean over waves); prologue {(s[:, :, 1] - s[:, :, 0]).mean():,.0f}")
prev = s[:, :, 1]
comp_sum = wait_sum = 0.0
for u, n in enumerate(names):
    mid, end = s[:, :, 2 + 2 * u], s[:, :, 3 + 2 * u]
    comp, wait = (mid - prev).mean(), (end - mid).mean()
    comp_sum += comp; wait_sum += wait
    print(f"  {n:6s} compute {comp:8,.0f}  (min {(mid - prev).min():6,d} max {(mid - prev).max():6,d})   DMA wait + barrier {wait:7,.0f}")
    prev = end
print(f"compute {comp_sum:,.0f}, waits {wait_sum:,.0f}")
