"""Phase shares of the reference-order Tucker pass from a -DTR_STAMPS build (timing-only: that build writes stamps instead of x_hat).
usage: NLML_HPE_LIB=exp_libs/trstamps.so python tools/td_ref_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
N = 4096
P = torch.from_numpy(synth.tucker_params(N)).to(dev)
X = torch.from_numpy(synth.features(N, 1404, 3)).to(dev)
for _ in range(3):
    err, xh = ops.tucker_objective(Wm, X, P, cp, return_xhat=True, order="reference")
torch.cuda.synchronize()
st = xh.cpu().numpy().view(np.uint64).reshape(-1)[: (N // 16) * 2 * 12 * 8].reshape(N // 16, 2, 12, 8).astype(np.int64)
names = ["factor table + ring prologue", "main loop", "x loads + d2 stores + barrier", "pairwise level 1", "level 2", "level 3"]
d = np.diff(st[..., :7], axis=-1)                      # [wg, pass, wave, phase]
tot = st[:, 1, :, 6] - st[:, 0, :, 0]
print("ticks per workgroup, both passes (wave mean): %.0f   gap between the passes: %.0f" % (tot.mean(), (st[:, 1, :, 0] - st[:, 0, :, 6]).mean()))
for i, n in enumerate(names):
    print(f"  {n:34s} {d[..., i].mean():10.0f} ticks per pass  ({100 * 2 * d[..., i].mean() / tot.mean():.1f} %)")
print("  main loop by wave (ticks, mean over workgroups and passes; waves w, w+4, w+8 share a SIMD):", " ".join("%d" % v for v in d[..., 1].mean(axis=(0, 1))))
print("  end of main loop relative to pass start, by wave:", " ".join("%d" % v for v in (st[..., 2] - st[..., 0].min(axis=2, keepdims=True)).mean(axis=(0, 1))))
G = N // 16
ks = xh.cpu().numpy().view(np.uint64).reshape(-1)[G * 192: G * 192 + G * 8].reshape(G, 8).astype(np.int64)
dt_ticks, dt_real = (ks[:, 3] - ks[:, 0]).mean(), (ks[:, 5] - ks[:, 4]).mean()
print("kernel body per workgroup: %.0f ticks = %.1f us (s_memrealtime, 100 MHz) -> %.2f ticks per ns;  f-vectors %.0f ticks, passes %.0f, end %.0f"
      % (dt_ticks, dt_real / 100.0, dt_ticks / (dt_real * 10.0), (ks[:, 1] - ks[:, 0]).mean(), (ks[:, 2] - ks[:, 1]).mean(), (ks[:, 3] - ks[:, 2]).mean()))
print("grid: first start to last end %.1f us (s_memrealtime); workgroup starts spread over %.1f us" % ((ks[:, 5].max() - ks[:, 4].min()) / 100.0, (ks[:, 4].max() - ks[:, 4].min()) / 100.0))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ops.tucker_objective(Wm, X, P, cp, order="reference")
torch.cuda.synchronize(); print("wall clock per launch (10 back to back): %.1f us" % ((time.perf_counter() - t0) / 10 * 1e6))
