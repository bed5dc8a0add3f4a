"""Shared plumbing of the root entry-point scripts (NLML_HPE_Test.py, TD_Inference.py,
generatePose_on_video.py, NLML_HPE_Model_Builder.py): config loading, model/artefact lookup with the
reference's relative paths, landmark-file input (this image has no cv2/MediaPipe, so inputs are
pre-extracted FaceMesh landmarks)."""
from __future__ import annotations

import os
import warnings

import numpy as np
import yaml

from . import _lib, synth
from .model import HIPPoseModel, load_model


def load_config(path: str) -> dict:
    with open(path, "r") as f:
        return yaml.safe_load(f)


def resolve_model(device, scripted_name: str = "models/combined_model_scripted.pth", model_dir: str = "models",
                  input_size: int | None = None, synthetic_seed: int | None = 0, mode=None) -> HIPPoseModel:
    """The scripted file the reference loads if present; else the per-network state dicts; the encoder
    falls back to SYNTHETIC weights (with a warning) because the reference ships no models/Encoder.pth.
    mode: "f16x2s" (default: strict-fast), "f32", "f16x2" (opt-in, 1.10x the reference's error) or "bf16" (model.HIPPoseModel); the NLML_HPE_MODE environment variable
    sets it for the entry-point scripts without touching their command lines."""
    mode = mode if mode is not None else os.environ.get("NLML_HPE_MODE", _lib.DEFAULT_MODE_NAME)
    if os.path.isfile(scripted_name):
        return load_model(scripted_name, device, mode=mode)
    try:
        return load_model(model_dir, device, mode=mode)
    except FileNotFoundError:
        if synthetic_seed is None:
            raise
        F = int(input_size or synth.F_REFERENCE)
        warnings.warn(f"{model_dir}/Encoder.pth not found: using SYNTHETIC encoder weights (seed {synthetic_seed}); "
                      "poses are numerically valid but not trained predictions")
        return load_model(model_dir, device, encoder_state_dict=synth.encoder_state_dict(F, synthetic_seed), mode=mode)


def load_landmarks(path: str) -> np.ndarray:
    """.npy / .npz (key 'landmarks') holding f32[468,3] or f32[N,468,3] raw FaceMesh coordinates."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if path.endswith(".npz"):
        arr = np.load(path)["landmarks"]
    elif path.endswith(".npy"):
        arr = np.load(path)
    else:
        raise ValueError(f"{path}: images need MediaPipe FaceMesh, which is not available here; "
                         "pass pre-extracted landmarks as .npy/.npz (f32[N,468,3])")
    arr = np.asarray(arr, dtype=np.float32)
    if arr.ndim == 2:
        arr = arr[None]
    if arr.shape[1:] != (468, 3):
        raise ValueError(f"landmarks must be [N,468,3], got {arr.shape}")
    return arr
