"""Artefact producers that are pure tensor algebra, on the device.

  heads_training_table   the (U_yaw, U_pitch, U_roll, angles) tables NLML_HPE_MLPHeadsTrainer.py:155-205 builds with two
                         nested Python loops over cosine(): rows = ground-truth angles of configs/config_MlpHeads.yaml,
                         columns = the optimised cosine rows of outputs/features/Trained_data.npz
  core_to_W              W = core x_5 U_feat (TD_main.py:232-238), the tensor TD_Tester.objective contracts

Training itself (the MLP fits, the Tucker decomposition, TD_Trainer) stays out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def angle_grid(min_bin: float, max_bin: float, interval: float) -> np.ndarray:
    """np.radians(np.arange(min, max, interval).astype(np.float32)) -- f32 radians (MLPHeadsTrainer.py:179-181)."""
    return np.radians(np.arange(min_bin, max_bin, interval).astype(np.float32))


def heads_training_table(config: dict, optimized: dict, device="cuda") -> dict:
    """-> {"yaw": (U f64[n,R] on device, angles f32[n]), "pitch": ..., "roll": ...}.

    config: the parsed configs/config_MlpHeads.yaml; optimized: {"yaw"|"pitch"|"roll": f64[R,4]} (optimized_* arrays)."""
    out = {}
    for name in ("yaw", "pitch", "roll"):
        b = config[f"{name}_bins"]
        ang = angle_grid(b["min_bin"], b["max_bin"], b["interval"])
        U = ops.cosine_table(torch.from_numpy(ang).to(device),
                             torch.from_numpy(np.ascontiguousarray(optimized[name], dtype=np.float64)).to(device))
        out[name] = (U, ang)
    return out


def core_to_W(core, U_feat, device="cuda") -> torch.Tensor:
    """core f32[5,3,3,3,R5], U_feat f32[M,R5] -> W f32[5,3,3,3,M] on the device."""
    c = torch.as_tensor(np.ascontiguousarray(core, dtype=np.float32)) if not torch.is_tensor(core) else core
    u = torch.as_tensor(np.ascontiguousarray(U_feat, dtype=np.float32)) if not torch.is_tensor(U_feat) else U_feat
    return ops.mode5_product(c.to(device), u.to(device))
