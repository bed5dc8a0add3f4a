"""Workload for rocprofv3: N calls of the 128-face-tile path (or the fused kernel with `fused`) on 65,536 raw-landmark faces.
usage: wide_workload.py [streamed|fused|ws] [calls] [batch] [f16x2s|bf16]   (the name is from the 128-face-tile path it was written for, deleted since)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nlml_hpe_amd import _lib, synth, weights

which = sys.argv[1] if len(sys.argv) > 1 else "fused"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
mode = _lib_mode = sys.argv[4] if len(sys.argv) > 4 else "f16x2s"
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.mode_from_name(mode))).to(dev)
L = _lib.lib()
raw = torch.from_numpy(synth.raw_landmarks(B, seed=1)).to(dev)
ws = torch.empty((L.nlml_encoder_heads_workspace_bytes(B, 1404),), dtype=torch.uint8, device=dev)
out = torch.empty((B, 3), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(calls):
    if which == "streamed":
        L.nlml_landmarks_to_pose_streamed(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st)
    elif which == "ws":
        L.nlml_landmarks_to_pose_ws(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st)
    else:
        L.nlml_landmarks_to_pose(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, st)
torch.cuda.synchronize()
print("done", which, calls, B)
