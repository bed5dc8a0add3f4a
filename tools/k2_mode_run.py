"""Run K2 in one mode a few times (profiling target).  usage: k2_mode_run.py <mode 0|1|2> <fused|features> [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nlml_hpe_amd import ops, synth, weights

mode = int(sys.argv[1]); path = sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, 0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, mode)).to(dev)
raw = torch.from_numpy(synth.raw_landmarks(B, 1)).to(dev)
feats = ops.normalize_ipd(raw, True)
fn = (lambda: ops.landmarks_to_pose(raw, blob, True)) if path == "fused" else (lambda: ops.encoder_heads_fwd(feats, blob, 1404))
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"mode {mode} {path} B={B}: {ms:.3f} ms  {B/ms*1e3/1e6:.2f} Mfaces/s", flush=True)
