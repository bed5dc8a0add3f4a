"""Latency of the layer-per-launch K2 path at 64 / 512 / 2,000 / 4,096 faces, both split-f16 modes (A/B of builds on ONE box:\nfor L in exp_libs/a.so exp_libs/b.so; do NLML_HPE_LIB=$L python tools/k2_small_ab.py; done).  Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nlml_hpe_amd import ops, synth, weights, _lib
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
def ms_of(fn, n=300, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name in ("f16x2", "f16x2s"):
    blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(1404, 0), heads, _lib.mode_from_name(name))).to(dev)
    for B in (64, 512, 2000, 4096):
        raw = torch.from_numpy(synth.raw_landmarks(B, seed=1)).to(dev)
        print(os.environ.get("NLML_HPE_LIB", "default"), name, B, "layered %.4f ms" % ms_of(lambda: ops.landmarks_to_pose_small(raw, blob, True)), flush=True)
