#!/usr/bin/env python3
"""Per-frame head pose with smoothing and axis end points for MANY concurrent video streams --
MI355X counterpart of the reference's generatePose_on_video.py (process_video, :128).

Capture, FaceMesh, drawing and XVID encoding (cv2 / MediaPipe) are unavailable here and out of
scope; the per-frame ARITHMETIC is kept: landmarks -> normalise -> encoder+heads (fused launch) ->
round(deg, 2) -> EMA (alpha 0.4) -> face centre with the 100-px jump gate -> axis end points, for S
streams per tick on the device (nlml_hpe_amd/video.py, csrc/video_post.hip).

    python generatePose_on_video.py --source clips.npz --save_output True --output_path poses.npz
        clips.npz: `landmarks` f32[T,S,468,3] (all-zero [468,3] = no face in that frame), optional `width`,`height`
    python generatePose_on_video.py --source synthetic --save_output False       # 64 streams x 90 ticks, 1080p
"""
from __future__ import annotations

import argparse
import time

import numpy as np
import torch

from nlml_hpe_amd import synth
from nlml_hpe_amd.entrypoints import resolve_model
from nlml_hpe_amd.video import VideoPoseTracker


def _synthetic_clips(T=90, S=64, seed=7):
    base = synth.raw_landmarks(S, seed=seed) * 0.2 + 0.4
    drift = 0.002 * synth.rng(seed, 5).standard_normal((T, S, 1, 3)).cumsum(axis=0)
    return (base[None] + drift).astype(np.float32)


def process_video(source, output_path, model, save_output, device="cuda:0"):
    if source == "synthetic":
        clips, width, height = _synthetic_clips(), 1920, 1080
    else:
        data = np.load(source)
        clips = np.asarray(data["landmarks"], np.float32)
        width = int(data["width"]) if "width" in data else 640      # fallbacks of the reference, :160-166
        height = int(data["height"]) if "height" in data else 480
    if clips.ndim != 4 or clips.shape[2:] != (468, 3):
        raise ValueError(f"landmarks must be [T,S,468,3], got {clips.shape}")
    T, S = clips.shape[:2]
    tracker = VideoPoseTracker(model, S, width, height)
    frames = torch.from_numpy(clips).to(device)
    lat = []
    out_sm, out_ep, out_valid = [], [], []
    torch.cuda.synchronize()
    t_start = time.time()
    for t in range(T):
        t0 = time.perf_counter()
        sm, centre, ep, valid = tracker.tick(frames[t])
        if save_output:
            out_sm.append(sm.clone()); out_ep.append(ep.clone()); out_valid.append(valid.clone())
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    total = time.time() - t_start
    lat = np.array(lat)
    print(f"average frame processing time = {lat.mean()}")
    print(f"{S} streams x {T} ticks: {S * T / total:,.0f} faces/s, tick latency p50 {np.percentile(lat, 50) * 1e3:.3f} ms "
          f"p99 {np.percentile(lat, 99) * 1e3:.3f} ms")
    if save_output:
        np.savez_compressed(output_path, smoothed_deg=torch.stack(out_sm).cpu().numpy(),
                            endpoints=torch.stack(out_ep).cpu().numpy(), valid=torch.stack(out_valid).cpu().numpy())
    return tracker


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--source", type=str, required=True)
    parser.add_argument("--save_output", type=lambda s: str(s).lower() in ("1", "true", "yes", "tr"), required=True)
    parser.add_argument("--output_path", type=str)
    parser.add_argument("--device", default="cuda:0")
    parser.add_argument("--mode", choices=["f16x2", "f32", "bf16"], default=None,
                        help="kernel mode (default: NLML_HPE_MODE or f16x2)")
    args = parser.parse_args()
    dev = torch.device(args.device)
    torch.cuda.set_device(dev)
    # the reference loads models/combined_model_scripted_prev.pth here (:289)
    mdl = resolve_model(dev, scripted_name="models/combined_model_scripted_prev.pth", mode=args.mode)
    mdl.eval()
    process_video(args.source, args.output_path, mdl, args.save_output, dev)
