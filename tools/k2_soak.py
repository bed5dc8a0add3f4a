"""Randomised soak of the K2 implementations that must agree bit for bit (GPU box): for <minutes> minutes draw a batch size (weighted towards
partial tiles and tile-count edges), a seed and a normalisation flag, and compare
  strict-fast: fused eight-wave kernel, twice (determinism)  ==  layer-per-launch path  ==  trunk + streamed-tail path  (pose, latent, validity)
  opt-in fast: fused four-wave kernel, twice                 ==  layer-per-launch path
  bf16 / f32 : fused kernel, twice (determinism)
and, every tenth draw, the full 65,536-face batch three times (four tiles per CU: anything one tile leaves behind for the next shows up here).
A few rows are zeroed ("no face") and one is made huge (f16 overflow -> the f32 re-evaluation launch) in some draws.
usage: k2_soak.py [minutes]      exit code 1 on the first difference (the draw is printed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import _lib, ops, synth, weights

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, 0)
blob = {m: torch.from_numpy(weights.pack_blob(sd, heads, _lib.mode_from_name(m))).to(dev) for m in ("f16x2s", "f16x2", "bf16", "f32")}
L = _lib.lib()
rng = np.random.default_rng(12345)


def streamed(raw, B, norm):
    return ops.landmarks_to_pose_streamed(raw, blob["f16x2s"], norm, return_latent=True, return_valid=True)


def same(a, b):
    return all(torch.equal(x.view(torch.uint8) if x.dtype != torch.bool else x, y.view(torch.uint8) if y.dtype != torch.bool else y) for x, y in zip(a, b))


def draw_B():
    k = rng.integers(0, 6)
    if k == 0: return int(rng.integers(1, 130))
    if k == 1: return int(64 * rng.integers(1, 70) + rng.integers(-1, 2))
    if k == 2: return int(128 * rng.integers(1, 40) + rng.integers(-2, 3))
    if k == 3: return int(rng.integers(4000, 4200))
    if k == 4: return int(rng.integers(130, 20000))
    return int(256 * 64 + rng.integers(-70, 70))


t_end = time.time() + 60 * minutes
n = 0
while time.time() < t_end:
    n += 1
    big = n % 10 == 0
    B = 65536 if big else max(1, draw_B())
    seed = int(rng.integers(0, 1 << 30))
    norm = bool(rng.integers(0, 2))
    raw_np = synth.raw_landmarks(B, seed)
    if rng.integers(0, 3) == 0:
        for r in rng.integers(0, B, size=min(B, 3)): raw_np[r] = 0.0
    if rng.integers(0, 4) == 0:
        raw_np[int(rng.integers(0, B))] *= 3.0e5        # leaves f16's range: the strict mode's re-evaluation launch, the fast mode's rescue
    raw = torch.from_numpy(raw_np).to(dev)
    tag = f"draw {n}: B={B} seed={seed} norm={norm}"
    ref = ops.landmarks_to_pose(raw, blob["f16x2s"], norm, return_latent=True, return_valid=True)
    for rep in range(3 if big else 1):
        again = ops.landmarks_to_pose(raw, blob["f16x2s"], norm, return_latent=True, return_valid=True)
        if not same(ref, again): print("DIFFERENT (strict fused, run to run)", tag); sys.exit(1)
    if not same(ref, streamed(raw, B, norm)): print("DIFFERENT (strict fused vs trunk + streamed tail)", tag); sys.exit(1)
    if not big:
        if B <= 16384:
            if not same(ref, ops.landmarks_to_pose_small(raw, blob["f16x2s"], norm, return_latent=True, return_valid=True)):
                print("DIFFERENT (strict fused vs layer-per-launch)", tag); sys.exit(1)
            f = ops.landmarks_to_pose(raw, blob["f16x2"], norm, return_latent=True, return_valid=True)
            if not same(f, ops.landmarks_to_pose(raw, blob["f16x2"], norm, return_latent=True, return_valid=True)):
                print("DIFFERENT (fast fused, run to run)", tag); sys.exit(1)
            if not same(f, ops.landmarks_to_pose_small(raw, blob["f16x2"], norm, return_latent=True, return_valid=True)):
                print("DIFFERENT (fast fused vs layer-per-launch)", tag); sys.exit(1)
    for m in ("bf16", "f32"):
        if m == "f32" and B > 20000 and not big: continue
        a = ops.landmarks_to_pose(raw, blob[m], norm, return_latent=True, return_valid=True)
        b = ops.landmarks_to_pose(raw, blob[m], norm, return_latent=True, return_valid=True)
        if not same(a, b): print(f"DIFFERENT ({m} fused, run to run)", tag); sys.exit(1)
    if n % 20 == 0: print(f"{n} draws, all identical ({time.time() - (t_end - 60 * minutes):.0f} s)", flush=True)
torch.cuda.synchronize()
print(f"soak done: {n} draws, every comparison bit-identical")
