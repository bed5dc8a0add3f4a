"""Development aid: eager vs hipGraph timing of the layer-per-launch path; run under rocprofv3 --kernel-trace for per-layer times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights, _lib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
heads = weights.load_head_state_dicts(os.path.join(R, "models"))
sd = synth.encoder_state_dict(1404, 0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_F16X2)).to(dev)
raw = torch.from_numpy(synth.raw_landmarks(B, 3)).to(dev)
fn = lambda: ops.landmarks_to_pose_small(raw, blob, True)
for _ in range(20): fn()
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): fn()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fn()
def t(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"B={B}: layered eager {t(fn):.1f} us, layered hipGraph {t(g.replay):.1f} us, fused {t(lambda: ops.landmarks_to_pose(raw, blob, True)):.1f} us", flush=True)
