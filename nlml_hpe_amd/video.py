"""Batched video post-processing: S concurrent streams, state resident on the device.

Mirror of the per-frame tail of process_video (generatePose_on_video.py:198-229): rounding, EMA
smoothing, face centre with the jump gate and the axis end points, for many streams per tick in one
HIP launch (csrc/video_post.hip).  Capture / drawing / encoding (cv2) are out of scope.
"""
from __future__ import annotations

import torch

from . import _lib, ops

ALPHA = 0.4              # generatePose_on_video.py:179
MAX_CENTER_JUMP = 100.0  # :136
AXIS_SIZE = 80.0         # :73


class VideoPoseTracker:
    def __init__(self, model, streams: int, frame_w: int, frame_h: int, alpha: float = ALPHA,
                 max_jump: float = MAX_CENTER_JUMP, size: float = AXIS_SIZE):
        self.model, self.S = model, int(streams)
        self.frame_w, self.frame_h = float(frame_w), float(frame_h)
        self.alpha, self.max_jump, self.size = float(alpha), float(max_jump), float(size)
        dev = model.device
        self.state = torch.zeros((self.S, 6), dtype=torch.float64, device=dev)
        self.smoothed = torch.zeros((self.S, 3), dtype=torch.float64, device=dev)
        self.centre = torch.zeros((self.S, 2), dtype=torch.float64, device=dev)
        self.endpoints = torch.zeros((self.S, 3, 2), dtype=torch.float64, device=dev)
        self.updated = torch.zeros((self.S,), dtype=torch.uint8, device=dev)   # 1 = the last tick was applied to the stream

    def post(self, pose_rad: torch.Tensor, raw: torch.Tensor, valid: torch.Tensor | None = None):
        """One tick given the model outputs: updates state; returns (smoothed deg, centre, endpoints) views."""
        ops._need_cuda(pose_rad, "pose_rad", torch.float32)
        ops._need_cuda(raw, "raw", torch.float32)
        if tuple(pose_rad.shape) != (self.S, 3) or tuple(raw.shape) != (self.S, 468, 3):
            raise ValueError(f"expected pose [{self.S},3] and landmarks [{self.S},468,3]")
        v8 = None
        if valid is not None:
            if tuple(valid.shape) != (self.S,):
                raise ValueError("valid: expected [S]")
            v8 = valid.to(torch.uint8).contiguous()
        pose_rad, raw = pose_rad.contiguous(), raw.contiguous()
        with ops._on_device_of(("state", self.state), ("pose_rad", pose_rad), ("raw", raw), ("valid", v8)) as stream:
            _lib.check(_lib.lib().nlml_video_post_ex(
                pose_rad.data_ptr(), raw.data_ptr(), v8.data_ptr() if v8 is not None else None,
                self.S, self.frame_w, self.frame_h, self.alpha, self.max_jump, self.size, self.state.data_ptr(),
                self.smoothed.data_ptr(), self.centre.data_ptr(), self.endpoints.data_ptr(), self.updated.data_ptr(), stream),
                "nlml_video_post_ex")
        return self.smoothed, self.centre, self.endpoints

    def tick(self, raw: torch.Tensor):
        """raw landmarks f32[S,468,3] of this tick (all-zero rows = no face) -> (smoothed, centre, endpoints, valid).
        valid[s] is True only where this tick was APPLIED to stream s: a face was found AND its pose is finite (a stream skipped for
        a NaN/Inf pose keeps its previous outputs and must not be saved as a new result; the reference raises on such a frame)."""
        # both launches through the compiled torch.ops (csrc/torch_ops.cpp): at 64 faces the Python wrappers' checks, allocations and
        # ctypes marshalling were ~2/3 of the tick's host time (tools/host_hop.py)
        m = self.model
        ops._need_cuda(raw, "raw", torch.float32)
        if tuple(raw.shape) != (self.S, 468, 3):
            raise ValueError(f"expected landmarks [{self.S},468,3]")
        ws = ops._small_workspace(self.S, ops.F_REF, m.device) if m._small(self.S) else None
        pose, valid = torch.ops.nlml_hpe.landmarks_to_pose_valid(raw, m.blob, True, ws)
        torch.ops.nlml_hpe.video_post(pose, raw, valid, self.frame_w, self.frame_h, self.alpha, self.max_jump, self.size,
                                      self.state, self.smoothed, self.centre, self.endpoints, self.updated)
        return self.smoothed, self.centre, self.endpoints, self.updated.bool()


class GraphedTick:
    """One tick (fused forward + post-processing, 2 launches) captured into a hipGraph and replayed.

    The launch functions allocate nothing and never synchronise (include/nlml_hpe.h), so a launch-bound
    loop -- 64 faces per tick is one tile on one CU; the cost is launch latency -- replays as one graph.
    Usage: g = GraphedTick(tracker); g.static_raw.copy_(raw_t); sm, centre, ep, valid = g.replay()
    """

    def __init__(self, tracker: VideoPoseTracker):
        self.tracker = tracker
        dev = tracker.model.device
        with torch.cuda.device(dev):                       # capture on the model's GPU, whatever the caller's current device
            self.static_raw = torch.zeros((tracker.S, 468, 3), dtype=torch.float32, device=dev)
            saved = tracker.state.clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                  # warm-up outside capture
                for _ in range(2):
                    tracker.tick(self.static_raw)
            torch.cuda.current_stream(dev).wait_stream(side)
            tracker.state.copy_(saved)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.outputs = tracker.tick(self.static_raw)
            tracker.state.copy_(saved)                     # the capture itself does not execute the tick

    def replay(self):
        with torch.cuda.device(self.tracker.model.device):
            self.graph.replay()
        return self.outputs
