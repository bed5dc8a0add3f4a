"""Fused kernel against the trunk + streamed-tail path by batch size (one process, alternating, HIP events): where the opt-in path starts to pay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nlml_hpe_amd import _lib, ops, synth, weights
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(1404, seed=0), heads, _lib.MODE_F16X2S)).to(dev)
base = torch.from_numpy(synth.raw_landmarks(65536, seed=1)).to(dev)


def t(fn, n, w):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for B in (8192, 16384, 24576, 32768, 49152, 65536, 131072, 262144):
    raw = torch.cat([base] * ((B + 65535) // 65536))[:B].contiguous()
    n = max(40, min(300, int(3e7 // B)))
    f, s = [], []
    for _ in range(2):
        f.append(t(lambda: ops.landmarks_to_pose(raw, blob, True), n, n // 3))
        s.append(t(lambda: ops.landmarks_to_pose_streamed(raw, blob, True), n, n // 3))
    fm, sm = min(f), min(s)
    print(f"B={B:7d}: fused {fm:.4f} ms ({B / fm / 1e3:6.1f} M faces/s)  streamed {sm:.4f} ms ({B / sm / 1e3:6.1f} M)  {100 * (fm / sm - 1):+.2f} %", flush=True)
