"""Fused kernel against the layer-per-launch path at 3,072 ... 16,384 faces, both split-f16 modes (the crossovers behind\nHIPPoseModel.small_batch_max).  Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nlml_hpe_amd import ops, synth, weights, _lib
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
def ms_of(fn, n=200, warm=60):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name in ("f16x2", "f16x2s"):
    blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(1404, 0), heads, _lib.mode_from_name(name))).to(dev)
    for B in (3072, 4096, 5120, 6144, 8192, 10240, 12288, 16384):
        raw = torch.from_numpy(synth.raw_landmarks(B, seed=1)).to(dev)
        print(name, B, "fused %.4f ms  layered %.4f ms" % (ms_of(lambda: ops.landmarks_to_pose(raw, blob, True)), ms_of(lambda: ops.landmarks_to_pose_small(raw, blob, True))), flush=True)
