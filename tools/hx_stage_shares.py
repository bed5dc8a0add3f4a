#!/usr/bin/env python3
"""Cycle-share breakdown of the split-f16 kernel from the -DHX_STAMPS diagnostic build (GPU box).
usage: NLML_HPE_LIB=$PWD/exp_libs/hx_stamps.so python tools/hx_stage_shares.py [fused|features]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights, _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
F, B = 1404, 65536
path = sys.argv[1] if len(sys.argv) > 1 else "features"
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(F, 0), heads, _lib.MODE_F16X2)).to(dev)
raw = torch.from_numpy(synth.raw_landmarks(B, 1)).to(dev)
x = ops.normalize_ipd(raw, True)
for _ in range(2):
    out, lat = (ops.landmarks_to_pose(raw, blob, True, return_latent=True) if path == "fused"
                else ops.encoder_heads_fwd(x, blob, F, return_latent=True))
torch.cuda.synchronize()
tiles = B // 64
st = lat.cpu().numpy().reshape(-1).view(np.uint64)[: tiles * 4 * 32].reshape(tiles, 4, 32).astype(np.int64)
names = ["E0 slabs (one pass)", "E0a store+bar", "E1a kloop", "E1a bar", "E0b store+bar", "E1b kloop", "E1b bar", "-",
         "h2 store+bar", "E2", "E3", "E4+E5", "H0 yaw", "H1 yaw", "H2-H4 yaw", "H0 pitch", "H1 pitch", "H2-H4 pitch",
         "H0 roll", "H1 roll", "H2-H4 roll"]
NS = len(names) + 1                                   # stamps 0 .. 21
d = np.diff(st[:, :, :NS], axis=2).astype(np.float64)
tot = (st[:, :, NS - 1] - st[:, :, 0]).astype(np.float64)
print(f"{path}: tiles {tiles}  mean cycles/tile (wave avg) {tot.mean():,.0f}  min {tot.min():,.0f} max {tot.max():,.0f}")
M = 32 * 3   # cycles per (nb, fb, K16) product = three 32-cycle MFMAs
ideal = {"E0 slabs (one pass)": 88 * 16 * M, "E1a kloop": 32 * 8 * M, "E1b kloop": 32 * 8 * M, "E2": 32 * 4 * M,
         "E3": 16 * 2 * M, "E4+E5": (8 + 4 * 2) * M}
for h in ("yaw", "pitch", "roll"):                    # one head, 64 faces: per wave H0 1x2, H1 8 steps x 2x2, H2 16 x 1x2, H3 8 x 1x1, H4 4 x 1x1
    ideal[f"H0 {h}"] = 2 * M
    ideal[f"H1 {h}"] = 8 * 4 * M
    ideal[f"H2-H4 {h}"] = (16 * 2 + 8 + 4) * M
for i, n in enumerate(names):
    m = d[:, :, i].mean()
    extra = f"  ideal MFMA {ideal[n]:,}  ({ideal[n]/m*100:.0f}% busy)" if n in ideal else ""
    print(f"{n:20s} {m:10,.0f} cyc  {m/tot.mean()*100:5.1f}%{extra}")
g0 = st[:, :, [12, 22, 23, 24, 25, 13, 26, 27, 28, 14]].astype(np.float64)          # the yaw head's first two stages in detail
for n, v in zip(["H0 pre issued", "H0 run", "H1 pre issued", "H0 store", "barrier", "H1 run", "H2 pre issued", "H1 store", "barrier"],
                np.diff(g0, axis=2).mean(axis=(0, 1))):
    print(f"   yaw detail: {n:16s} {v:8,.0f} cyc")
print(f"   E0b detail: store {np.mean(st[:, :, 29] - st[:, :, 4]):,.0f} cyc, barrier wait {np.mean(st[:, :, 5] - st[:, :, 29]):,.0f} cyc "
      f"(per wave: store {[int(v) for v in np.mean(st[:, :, 29] - st[:, :, 4], axis=0)]}, wait {[int(v) for v in np.mean(st[:, :, 5] - st[:, :, 29], axis=0)]})")
wall = (st[:, :, 31] - st[:, :, 30]).astype(np.float64)      # 100 MHz ticks
print(f"core clock during a tile: {tot.mean() / (wall.mean() * 10.0):.3f} GHz (s_memtime / s_memrealtime); tile wall time {wall.mean()*0.01:.1f} us; "
      f"first start {st[:, :, 30].min()}, last end {st[:, :, 31].max()} -> kernel span {(st[:, :, 31].max() - st[:, :, 30].min())*0.01:.1f} us")
print(f"sum of ideal MFMA cycles {sum(ideal.values()):,} = {sum(ideal.values())/tot.mean()*100:.1f}% of the tile")
