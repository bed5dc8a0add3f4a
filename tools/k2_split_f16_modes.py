"""The two split-f16 K2 modes side by side (development aid): rate of the fused and of the layer-per-launch path at
4,096 / 16,384 / 65,536 faces, and the FX3c statistics (distance from the f64 result at the reference's operating range).
usage (GPU box): python tools/k2_split_f16_modes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixture_models
from nlml_hpe_amd import ops, synth, weights, _lib
from oracle import encoder_heads as EH

dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))


def ms_of(fn, n=50, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


sd0 = synth.encoder_state_dict(1404, 0)
g, sd, x = fixture_models.fx3c(os.path.join(ROOT, "tests", "golden"))
truth = EH.forward_numpy(x, EH.Params(sd, heads), np.float64)
print("reference (torch f32, batched)", fixture_models.error_stats(g["rad"], truth), flush=True)
for name in ("f16x2", "f16x2s"):
    mode = _lib.mode_from_name(name)
    blob = torch.from_numpy(weights.pack_blob(sd0, heads, mode)).to(dev)
    for B in (4096, 16384, 65536):
        xt = torch.from_numpy(synth.features(B, 1404, 1)).to(dev)
        f = ms_of(lambda: ops.encoder_heads_fwd(xt, blob, 1404))
        l = ms_of(lambda: ops.encoder_heads_fwd_small(xt, blob, 1404))
        print(f"{name}: B={B}: fused {f:.3f} ms ({B / f / 1e3:.1f} M faces/s)   layered {l:.3f} ms ({B / l / 1e3:.1f} M faces/s)", flush=True)
    bl = torch.from_numpy(weights.pack_blob(sd, heads, mode)).to(dev)
    xt = torch.from_numpy(x).to(dev)
    fused = ops.encoder_heads_fwd(xt, bl, 1404).cpu().numpy()
    small = np.concatenate([ops.encoder_heads_fwd_small(xt[i:i + 4096].contiguous(), bl, 1404).cpu().numpy() for i in range(0, 16384, 4096)])
    print(f"{name}: FX3c", fixture_models.error_stats(fused, truth), " layered == fused:", np.array_equal(small, fused), flush=True)
