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

Several GPUs (BASELINE config 5, "8 x MI355X, 64 concurrent streams"): started under torch.distributed.run (WORLD_SIZE ranks, one
per GPU) the STREAMS are sharded over the ranks as contiguous blocks (nlml_hpe_amd.distributed.shard_bounds): a stream's state
(EMA, previous centre) lives on one GPU for its whole life and a tick needs no exchange between ranks -- streams are independent
(generatePose_on_video.py:128 processes one video per process).  Only the saved outputs are collated, once, at the end.  One GPU
carries 64 streams at 0.3 % load (1,920 faces/s offered, >600,000 sustained), so sharding buys latency head-room, not throughput.
"""
from __future__ import annotations

import argparse
import time

import os

import numpy as np
import torch

from nlml_hpe_amd import synth
from nlml_hpe_amd.distributed import shard_bounds
from nlml_hpe_amd.entrypoints import resolve_model
from nlml_hpe_amd.video import VideoPoseTracker


def _synthetic_clips(T=90, S=64, seed=7):
    base = synth.raw_landmarks(S, seed=seed) * 0.2 + 0.4
    drift = 0.002 * synth.rng(seed, 5).standard_normal((T, S, 1, 3)).cumsum(axis=0)
    return (base[None] + drift).astype(np.float32)


def _collate(local: torch.Tensor, per: int, S: int, world: int) -> torch.Tensor:
    """[T, streams of this rank, ...] of every rank -> [T, S, ...] (ranks hold contiguous stream blocks, padded to `per`)."""
    import torch.distributed as dist
    if dist.get_backend() == "gloo":                          # one-GPU rehearsal of the N > 1 path: gloo collates on the host
        local = local.cpu()
    T = local.shape[0]
    pad = torch.zeros((T, per) + tuple(local.shape[2:]), dtype=local.dtype, device=local.device)
    pad[:, :local.shape[1]] = local
    out = torch.empty((world * T, per) + tuple(local.shape[2:]), dtype=local.dtype, device=local.device)   # rank-major
    dist.all_gather_into_tensor(out, pad.contiguous())
    out = out.reshape((world, T, per) + tuple(local.shape[2:]))
    return out.transpose(0, 1).reshape((T, world * per) + tuple(local.shape[2:]))[:, :S]


def process_video(source, output_path, model, save_output, device="cuda:0", world=1, rank=0):
    if source == "synthetic":
        clips, width, height = _synthetic_clips(), 1920, 1080
    else:
        data = np.load(source)
        clips = np.asarray(data["landmarks"], np.float32)
        width = int(data["width"]) if "width" in data else 640      # fallbacks of the reference, :160-166
        height = int(data["height"]) if "height" in data else 480
    if clips.ndim != 4 or clips.shape[2:] != (468, 3):
        raise ValueError(f"landmarks must be [T,S,468,3], got {clips.shape}")
    T, S_all = clips.shape[:2]
    # decided from (S_all, world) alone, so EVERY rank takes the same exit before any collective: a rank that left alone here would
    # leave the others blocked in the collation's all-gather
    empty = [r for r in range(world) if shard_bounds(S_all, world, r)[0] >= shard_bounds(S_all, world, r)[1]]
    if empty:
        raise SystemExit(f"{S_all} streams over {world} ranks leaves ranks {empty} without a stream: use at most {S_all} ranks")
    s0, s1, per = shard_bounds(S_all, world, rank)            # this rank's contiguous block of streams
    clips = clips[:, s0:s1]
    S = s1 - s0
    tracker = VideoPoseTracker(model, S, width, height)
    frames = torch.from_numpy(np.ascontiguousarray(clips)).to(device)
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
    tag = f"[rank {rank}/{world}, streams {s0}..{s1 - 1}] " if world > 1 else ""
    print(f"{tag}average frame processing time = {lat.mean()}")
    print(f"{tag}{S} streams x {T} ticks: {S * T / total:,.0f} faces/s, tick latency p50 {np.percentile(lat, 50) * 1e3:.3f} ms "
          f"p99 {np.percentile(lat, 99) * 1e3:.3f} ms")
    if save_output:
        sm, ep, va = torch.stack(out_sm), torch.stack(out_ep), torch.stack(out_valid).to(torch.uint8)
        if world > 1:                                         # the one collective of this entry point: collate the outputs
            sm, ep, va = (_collate(t, per, S_all, world) for t in (sm, ep, va))
        if rank == 0:
            np.savez_compressed(output_path, smoothed_deg=sm.cpu().numpy(), endpoints=ep.cpu().numpy(),
                                valid=va.cpu().numpy().astype(bool))
    return tracker


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--source", type=str, required=True)
    parser.add_argument("--save_output", type=lambda s: str(s).lower() in ("1", "true", "yes", "tr"), required=True)
    parser.add_argument("--output_path", type=str)
    parser.add_argument("--device", default="cuda:0")
    parser.add_argument("--mode", choices=["f16x2", "f16x2s", "f32", "bf16"], default=None,
                        help="kernel mode (default: NLML_HPE_MODE or f16x2s = strict-fast; f32 = strict parity; f16x2 = opt-in, 1.10x the reference's error; bf16 = throughput only)")
    args = parser.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:                                             # one rank per GPU; RCCL ("nccl") unless rehearsing on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rehearsal = os.environ.get("NLML_BENCH_REHEARSAL") == "1"
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        args.device = f"cuda:{local}"
        dist.init_process_group(backend="gloo" if rehearsal else "nccl")
    dev = torch.device(args.device)
    torch.cuda.set_device(dev)
    # the reference loads models/combined_model_scripted_prev.pth here (:289)
    mdl = resolve_model(dev, scripted_name="models/combined_model_scripted_prev.pth", mode=args.mode)
    mdl.eval()
    process_video(args.source, args.output_path, mdl, args.save_output, dev, world, rank)
    if world > 1:
        torch.distributed.destroy_process_group()
