"""Ad-hoc GPU timing of the kernels (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights

dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts("models")
for F, B in ((1404, 65536), (136, 65536)):
    sd = synth.encoder_state_dict(F, 0)
    blob = torch.from_numpy(weights.pack_blob(sd, heads)).to(dev)
    x = torch.from_numpy(synth.features(B, F, 1)).to(dev)
    for _ in range(3):
        ops.encoder_heads_fwd(x, blob, F)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.encoder_heads_fwd(x, blob, F)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = {1404: 4714240, 136: 2117376}[F]
    print(f"K2 F={F} B={B}: {ms:.3f} ms  {B/ms*1e3/1e6:.2f} Mfaces/s  {B*flops/ms/1e9:.1f} TFLOP/s ({B*flops/ms/1e9/157.3*100:.1f}% of f32 MFMA peak)")
raw = torch.from_numpy(synth.raw_landmarks(65536, 1)).to(dev)
for _ in range(3): ops.normalize_ipd(raw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.normalize_ipd(raw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"K1 B=65536: {ms*1e3:.1f} us  {65536*11232/ms/1e6:.1f} GB/s")
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
N = 4096
P = torch.from_numpy(synth.tucker_params(N)).to(dev)
X = torch.from_numpy(synth.features(N, 1404, 3)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
for _ in range(3): ops.tucker_objective(Wm, X, P, cp, order="fast")
torch.cuda.synchronize()
e0.record()
for _ in range(20): ops.tucker_objective(Wm, X, P, cp, order="fast")
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"K3 N={N}: {ms*1e3:.1f} us  {N/ms*1e3/1e6:.2f} Mevals/s  {N*383.7e3/ms/1e9:.2f} TFLOP/s f64")
from nlml_hpe_amd import _lib
for F, B in ((1404, 65536), (136, 65536)):
    sd = synth.encoder_state_dict(F, 0)
    blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_BF16)).to(dev)
    x = torch.from_numpy(synth.features(B, F, 1)).to(dev)
    for _ in range(3):
        ops.encoder_heads_fwd(x, blob, F)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.encoder_heads_fwd(x, blob, F)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = {1404: 4714240, 136: 2117376}[F]
    print(f"K2-bf16 F={F} B={B}: {ms:.3f} ms  {B/ms*1e3/1e6:.2f} Mfaces/s  {B*flops/ms/1e9:.1f} TFLOP/s ({B*flops/ms/1e9/2500*100:.1f}% of bf16 MFMA peak)")
F = 1404
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(F, 0), heads)).to(dev)
for B in (64, 2000, 8192, 16384, 16448, 32768):
    x = torch.from_numpy(synth.features(B, F, 1)).to(dev)
    for _ in range(3):
        ops.encoder_heads_fwd(x, blob, F)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.encoder_heads_fwd(x, blob, F)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"K2 small-batch F={F} B={B}: {ms:.3f} ms  {B/ms*1e3/1e6:.2f} Mfaces/s")
