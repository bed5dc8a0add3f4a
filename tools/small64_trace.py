"""Workload for a kernel trace of the layer-per-launch path at one size (default 64 faces, strict mode): 200 calls.
rocprofv3 --kernel-trace --stats -d gpurun_out/x -- python3 tools/small64_trace.py [faces] [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nlml_hpe_amd import _lib, ops, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = _lib.MODE_NAMES[sys.argv[2]] if len(sys.argv) > 2 else _lib.MODE_F16X2S
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(1404, seed=0), heads, mode)).to(dev)
raw = torch.from_numpy(synth.raw_landmarks(B, seed=7)).to(dev)
for _ in range(200):
    ops.landmarks_to_pose_small(raw, blob, True)
torch.cuda.synchronize()
