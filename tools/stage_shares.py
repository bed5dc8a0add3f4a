#!/usr/bin/env python3
"""Cycle-share breakdown of the fused kernel from the diagnostic build's s_memtime stamps (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
dev = torch.device("cuda:0")
F, B = 1404, 65536
heads = weights.load_head_state_dicts("models")
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(F, 0), heads)).to(dev)
x = torch.from_numpy(synth.features(B, F, 1)).to(dev)
for _ in range(2):
    out, lat, pre, st = ops.encoder_heads_fwd_debug(x, blob, F, want_stamps=True)
torch.cuda.synchronize()
st = st.cpu().numpy().astype(np.int64)            # [tiles, 4 waves, 16]
names = ["E0a kloop", "E0a store+bar", "E1a kloop", "E1a bar", "E0b kloop", "E0b store+bar", "E1b kloop", "E1b bar",
         "valid+h2 store+bar", "E2", "E3", "E4+E5", "heads (2 x 32 faces)"]
d = np.diff(st[:, :, :14], axis=2).astype(np.float64)   # [tiles,4,13]
tot = (st[:, :, 13] - st[:, :, 0]).astype(np.float64)
print(f"tiles {st.shape[0]} (64 faces)  mean cycles/tile (wave avg) {tot.mean():,.0f}  min {tot.min():,.0f} max {tot.max():,.0f}")
M = 64  # cycles per MFMA
ideal = {"E0a kloop": 176*32*M, "E0b kloop": 176*32*M, "E1a kloop": 64*32*M, "E1b kloop": 64*32*M, "E2": 64*16*M, "E3": 32*8*M,
         "E4+E5": (16*4 + 8*4)*M, "heads (2 x 32 faces)": 2*(3*4 + 3*16*8 + 3*32*4 + 2*16*4 + 8*4)*M}
for i, n in enumerate(names):
    m = d[:, :, i].mean()
    extra = f"  ideal MFMA {ideal[n]:,}  ({ideal[n]/m*100:.0f}% busy)" if n in ideal else ""
    print(f"{n:22s} {m:10,.0f} cyc  {m/tot.mean()*100:5.1f}%{extra}")
print("per-wave totals:", [f"{tot[:, w].mean():,.0f}" for w in range(4)])
print(f"sum of ideal MFMA cycles {sum(ideal.values()):,} = {sum(ideal.values())/tot.mean()*100:.1f}% of the tile")
