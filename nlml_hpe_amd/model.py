"""Drop-in callable for the reference's scripted model.

The reference obtains its model with
    model = torch.jit.load("models/combined_model_scripted.pth", map_location=device); model.to(device); model.eval()
(NLML_HPE_Test.py:217-219, generatePose_on_video.py:289-290) and calls ``model(x)`` with x f32[B,1404]
under no_grad, getting three [B,1] tensors in RADIANS (NLML_HPE_Model_Builder.py:115-126).
``load_model`` returns an object with the same call contract whose forward is ONE fused HIP launch.
"""
from __future__ import annotations

import os

import torch

from . import _lib, ops, weights


class HIPPoseModel:
    """CombinedAnglePredictionModel (Model_Builder.py:107-126) on the fused gfx950 kernel."""

    def __init__(self, encoder_sd: dict, head_sds: dict, device="cuda", mode=_lib.DEFAULT_MODE):
        """mode (int constant or name):
          _lib.MODE_F16X2S "f16x2s" (DEFAULT) the strict-fast mode, on the f16 matrix cores: every f32 operand as two f16 pieces,
                                   the small products of each K step in accumulators of their own (layers 0 to 2): at the
                                   reference's operating range (poses to +-60 deg, FX3c) 1.50e-5 / 4.65e-5 / 9.2e-5 deg from the
                                   exact result = 0.88x the PINNED reference's distance / 1.14x torch-f32's on the GPU box's host;
                                   0.03-0.07 % of the faces differ from the reference's batched output by more than 1e-4 deg
                                   (f32 mode: 0.002-0.012 %).  A face whose activations leave f16's range has its tile
                                   re-evaluated on the f32 matrix cores by a SECOND launch behind the kernel, from the f32 image
                                   inside the strict blob (which is ~2x the size of the other modes'): no input-range limit;
          _lib.MODE_F16X2 "f16x2"  OPT-IN fast mode, 1.10x THE REFERENCE'S ERROR: the same operands on single accumulators
                                   where registers are short: 1.86e-5 / 5.9e-5 / 1.22e-4 deg from the exact result in
                                   p50 / p99 / max at the operating range = 1.10 / 1.08 / 1.24x the reference's own distance,
                                   0.024 % of the faces beyond 1e-4 deg; same range behaviour as f16x2s;
          _lib.MODE_F32   "f32"    the strict parity mode, on the f32 matrix cores: layers 0 to 3 summed in blocks of
                                   128 k, bit-identical to the C oracle's order 2; at the operating range 1.25e-5 / 4.0e-5 /
                                   8.7e-5 deg from the exact result -- no further out than the reference itself; no range limit;
          _lib.MODE_BF16  "bf16"   throughput mode (bf16 weights/activations, ~0.1 deg from the reference -- never a
                                   parity result)."""
        self.input_size = weights.validate_shapes(encoder_sd, head_sds)
        mode = _lib.mode_from_name(mode)
        self.mode = mode
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.NlmlError("HIPPoseModel runs on the GPU only; there is no CPU fallback")
        self._blob_host = torch.from_numpy(weights.pack_blob(encoder_sd, head_sds, mode))
        self.blob = self._blob_host.to(self.device)

    # -- nn.Module-shaped surface the reference's entry points touch -------------------------
    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.NlmlError("HIPPoseModel runs on the GPU only; there is no CPU fallback")
        if device != self.device:
            self.device = device
            self.blob = self._blob_host.to(device)
        return self

    def __call__(self, x: torch.Tensor):
        """x f32[B,F] (or [F]) -> (yaw[B,1], pitch[B,1], roll[B,1]) radians, like the scripted model."""
        out = self.forward_packed(x)
        return out[:, 0:1], out[:, 1:2], out[:, 2:3]

    forward = __call__

    # -- batched surface this build adds ------------------------------------------------------
    def forward_packed(self, x: torch.Tensor, return_latent: bool = False, return_valid: bool = False):
        """x f32[B,F] -> f32[B,3] radians (+ latent [B,9], + valid mask: row not all-zero)."""
        if x.dim() == 1:
            x = x.unsqueeze(0)
        fwd = ops.encoder_heads_fwd_small if self._small(x.shape[0]) else ops.encoder_heads_fwd
        return fwd(x.to(self.device, torch.float32), self.blob, self.input_size,
                   return_latent=return_latent, return_valid=return_valid)

    SMALL_BATCH_MAX = 4096          # NLML_MODE_F16X2: above this the fused kernel wins (5,120 faces: 0.140 against 0.159 ms)
    SMALL_BATCH_MAX_STRICT = 4096   # NLML_MODE_F16X2S: with the eight-wave fused kernel the same crossover (5,120 faces: 0.152 against 0.159 ms
                                    # layered; 4,096: 0.152 against 0.123; round 3's four-wave kernel lost up to 8,192)

    @classmethod
    def small_batch_max(cls, mode) -> int:
        """Largest batch the layer-per-launch path takes in `mode` (0 for the modes that have no such path); measured crossovers,
        tools/k2_crossover.py."""
        mode = _lib.mode_from_name(mode)
        return {_lib.MODE_F16X2: cls.SMALL_BATCH_MAX, _lib.MODE_F16X2S: cls.SMALL_BATCH_MAX_STRICT}.get(mode, 0)

    def _small(self, B: int) -> bool:
        """Split-f16 modes, up to small_batch_max(mode) faces: the layer-per-launch path (same bits as the fused kernel, 0.06-0.13 ms
        instead of 0.14-0.20 ms because a handful of 64-face tiles cannot fill 256 CUs with one CU per tile)."""
        return 0 < B <= self.small_batch_max(self.mode)

    def from_landmarks(self, raw: torch.Tensor, normalize: bool = True, return_latent: bool = False,
                       return_valid: bool = False):
        """raw FaceMesh landmarks f32[B,468,3] -> f32[B,3] radians, normalisation fused into the launch."""
        if self.input_size != ops.F_REF:
            raise ValueError("from_landmarks needs the reference input width 1404")
        fwd = ops.landmarks_to_pose_small if self._small(raw.shape[0]) else ops.landmarks_to_pose
        return fwd(raw.to(self.device, torch.float32), self.blob, normalize,
                   return_latent=return_latent, return_valid=return_valid)


def load_model(path_or_dir: str = "models", device=None, encoder_state_dict: dict | None = None,
               mode=_lib.DEFAULT_MODE) -> HIPPoseModel:
    """Build the HIP model from the reference's artefact layout.

    path_or_dir: a TorchScript file saved by NLML_HPE_Model_Builder.py (:222-223), or a directory holding
    Encoder.pth and {yaw,pitch,roll}_network.pth.  The reference checkout ships no Encoder.pth; pass
    ``encoder_state_dict`` (keys encoder.N.weight/bias) to supply encoder weights explicitly.
    """
    device = torch.device(device if device is not None else "cuda")
    if os.path.isfile(path_or_dir):
        scripted = torch.jit.load(path_or_dir, map_location="cpu")
        enc, heads = weights.split_scripted_state_dict(scripted.state_dict())
        if encoder_state_dict is not None:
            enc = encoder_state_dict
    else:
        heads = weights.load_head_state_dicts(path_or_dir)
        enc = encoder_state_dict if encoder_state_dict is not None else weights.load_encoder_state_dict(path_or_dir)
    return HIPPoseModel(enc, heads, device=device, mode=mode)
