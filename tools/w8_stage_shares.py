#!/usr/bin/env python3
"""Cycle shares of the eight-wave strict kernel from the -DHX_STAMPS diagnostic build (GPU box).
usage: NLML_HPE_LIB=$PWD/exp_libs/w8_stamps.so python tools/w8_stage_shares.py [fused|features]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights, _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
F, B = 1404, 65536
path = sys.argv[1] if len(sys.argv) > 1 else "fused"
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(F, 0), heads, _lib.MODE_F16X2S)).to(dev)
raw = torch.from_numpy(synth.raw_landmarks(B, 1)).to(dev)
x = ops.normalize_ipd(raw, True)
for _ in range(60):   # warm clocks
    ops.landmarks_to_pose(raw, blob, True)
for _ in range(2):
    out, lat = (ops.landmarks_to_pose(raw, blob, True, return_latent=True) if path == "fused"
                else ops.encoder_heads_fwd(x, blob, F, return_latent=True))
torch.cuda.synchronize()
tiles = B // 64
u = lat.cpu().numpy().reshape(-1).view(np.uint64)
tr = u[: tiles * 8 * 16].reshape(tiles, 8, 16).astype(np.int64)
tl = u[tiles * 8 * 16: tiles * 8 * 16 + tiles * 4 * 32].reshape(tiles, 4, 32).astype(np.int64)
M = 32 * 3
names = ["E0 pass A", "unpark+store+bars", "E1a kloop", "bar+park", "E0 pass B", "unpark+store+bars", "E1b kloop", "bar", "h2 store+bar", "E2 kloop", "E3 pre+bar+store+bar"]
ideal = {0: 88 * 4 * M * 2, 2: 32 * 4 * M * 2, 4: 88 * 4 * M * 2, 6: 32 * 4 * M * 2, 9: 32 * 2 * M * 2}   # x2: two waves share a SIMD's matrix pipe
d = np.diff(tr[:, :, :12], axis=2).astype(np.float64)
trunk = (tr[:, :, 11] - tr[:, :, 0]).astype(np.float64)
print(f"{path}: trunk cycles per tile (wave avg) {trunk.mean():,.0f}")
for i, n in enumerate(names):
    m = d[:, :, i].mean()
    # busy = matrix-pipe time of the SIMD's two waves over the SLOWER wave class's time (waves w and w + 4 share a SIMD and the older
    # one is served first: where a loop has no barrier inside, the stage lasts as long as waves 4-7 do, not as long as the average)
    slow = d[:, :, i].mean(axis=0).max()
    e = f"   MFMA-pipe time of the SIMD's two waves {ideal[i]:,} ({ideal[i] / slow * 100:.0f}% busy against the slower wave class)" if i in ideal else ""
    print(f"  {n:24s} {m:10,.0f}  {m / trunk.mean() * 100:5.1f}%{e}   per wave {[int(v) for v in d[:, :, i].mean(axis=0)]}")
tail = (tl[:, :, 21] - tr[:, :4, 11]).astype(np.float64)
print(f"tail (waves 0-3; E3 .. heads) {tail.mean():,.0f} cycles;  tile = {trunk.mean() + tail.mean():,.0f}")
tn = {11: "E3", 12: "E4+E5", 13: "H0 yaw", 14: "H1 yaw", 15: "H2-4 yaw", 16: "H0 pitch", 17: "H1 pitch", 18: "H2-4 pitch", 19: "H0 roll", 20: "H1 roll", 21: "H2-4 roll"}
prev = tr[:, :4, 11]
for i in range(11, 22):
    print(f"  {tn[i]:12s} {np.mean(tl[:, :, i] - prev):10,.0f}")
    prev = tl[:, :, i]
wall = (tr[:, :, 15] - tr[:, :, 14]).astype(np.float64)
print(f"clock during the trunk: {trunk.mean() / (wall.mean() * 10.0):.3f} GHz; trunk wall {wall.mean() * 0.01:.1f} us")
byts = (5.77e6 + 2 * 64 * 1404 * 4) / 2
print(f"layer-0 intake per pass {byts/1e6:.2f} MB -> {byts / d[:, :, 0].mean():.1f} / {byts / d[:, :, 4].mean():.1f} B/clk (pass A / B)")
