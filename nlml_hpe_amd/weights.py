"""Artefact loading in the reference's file layout, and packing to the kernel's blob.

File-layout contract kept from the reference:
  models/Encoder.pth                    keys encoder.{0,2,4,6,8,10}.{weight,bias}
                                        (NLML_HPE_Model_Builder.py:33-53,202) -- ABSENT from the
                                        reference checkout (.MISSING_LARGE_BLOBS); when missing, the
                                        caller must pass synthetic weights explicitly
  models/{yaw,pitch,roll}_network.pth   keys model.{0,2,4,6,8}.{weight,bias} (:76-92,212-214)
  models/combined_model_scripted.pth    TorchScript file with keys encoder.encoder.N.*,
                                        {yaw,pitch,roll}_network.model.N.* (:222-223)
  outputs/features/Trained_data.npz     optimized_{yaw,pitch,roll} f64(3,4), CoreTensor, W
  outputs/features/Factor_Matrices.npz  U_yaw (11,3) U_pitch (9,3) U_roll (7,3) U_id (1620,5)
  configs/config_EncoderTrainer.yaml    input_size (F)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

ENCODER_IDX = (0, 2, 4, 6, 8, 10)
HEAD_IDX = (0, 2, 4, 6, 8)
HEAD_NAMES = ("yaw", "pitch", "roll")
ENCODER_OUT = (1024, 512, 256, 128, 64, 9)
HEAD_SHAPES = ((128, 3), (256, 128), (128, 256), (64, 128), (1, 64))


def _np32(t) -> np.ndarray:
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def load_head_state_dicts(model_dir: str = "models") -> dict:
    import torch
    return {n: torch.load(os.path.join(model_dir, f"{n}_network.pth"), map_location="cpu", weights_only=True)
            for n in HEAD_NAMES}


def load_encoder_state_dict(model_dir: str = "models") -> dict:
    import torch
    path = os.path.join(model_dir, "Encoder.pth")
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found (the reference repo does not ship it); pass encoder weights explicitly, "
            "e.g. nlml_hpe_amd.synth.encoder_state_dict(F, seed) for synthetic ones")
    return torch.load(path, map_location="cpu", weights_only=True)


def split_scripted_state_dict(sd: dict):
    """state_dict of the TorchScript combined model -> (encoder_sd, {head: sd}) in per-file key form."""
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.encoder.")}
    heads = {n: {k[len(f"{n}_network."):]: v for k, v in sd.items() if k.startswith(f"{n}_network.")} for n in HEAD_NAMES}
    return enc, heads


def validate_shapes(encoder_sd: dict, head_sds: dict) -> int:
    """Checks the architecture is the reference's; returns F."""
    F = int(np.shape(encoder_sd["encoder.0.weight"])[1])
    fan_in = F
    for i, n_out in zip(ENCODER_IDX, ENCODER_OUT):
        w, b = np.shape(encoder_sd[f"encoder.{i}.weight"]), np.shape(encoder_sd[f"encoder.{i}.bias"])
        if tuple(w) != (n_out, fan_in) or tuple(b) != (n_out,):
            raise ValueError(f"encoder.{i}: expected ({n_out},{fan_in}), got {w}/{b}: only the reference architecture "
                             "(NLML_HPE_Model_Builder.py:33-53, latent 3x(1,3)) is supported")
        fan_in = n_out
    for n in HEAD_NAMES:
        for i, shp in zip(HEAD_IDX, HEAD_SHAPES):
            w = tuple(np.shape(head_sds[n][f"model.{i}.weight"]))
            if w != shp:
                raise ValueError(f"{n}_network model.{i}.weight: expected {shp}, got {w} (Model_Builder.py:76-92)")
    return F


def pack_blob(encoder_sd: dict, head_sds: dict, mode: int = _lib.MODE_F32) -> np.ndarray:
    """Pack into the MFMA fragment-order blob (layout: nlml_hpe_amd/csrc/layout.h). Host uint8 array."""
    F = validate_shapes(encoder_sd, head_sds)
    L = _lib.lib()
    nbytes = L.nlml_encoder_heads_packed_bytes(F, mode)
    if nbytes == 0:
        raise _lib.NlmlError(f"unsupported (F={F}, mode={mode})")
    keep = []

    def ptr(a):
        a = _np32(a)
        keep.append(a)
        return a.ctypes.data_as(C.c_void_p).value

    enc_w = (C.c_void_p * 6)(*[ptr(encoder_sd[f"encoder.{i}.weight"]) for i in ENCODER_IDX])
    enc_b = (C.c_void_p * 6)(*[ptr(encoder_sd[f"encoder.{i}.bias"]) for i in ENCODER_IDX])
    head_w = (C.c_void_p * 15)(*[ptr(head_sds[n][f"model.{i}.weight"]) for n in HEAD_NAMES for i in HEAD_IDX])
    head_b = (C.c_void_p * 15)(*[ptr(head_sds[n][f"model.{i}.bias"]) for n in HEAD_NAMES for i in HEAD_IDX])
    blob = np.zeros(nbytes, dtype=np.uint8)
    _lib.check(L.nlml_encoder_heads_pack(F, mode, enc_w, enc_b, head_w, head_b, blob.ctypes.data_as(C.c_void_p), nbytes),
               "nlml_encoder_heads_pack")
    return blob


def load_tucker_artefacts(feature_dir: str = "outputs/features") -> dict:
    """W, cosine parameters and factor matrices exactly as TD_Inference.py:40-51 reads them."""
    td = np.load(os.path.join(feature_dir, "Trained_data.npz"))
    fm = np.load(os.path.join(feature_dir, "Factor_Matrices.npz"))
    return {
        "W": td["W"], "CoreTensor": td["CoreTensor"],
        "optimized_yaw": td["optimized_yaw"], "optimized_pitch": td["optimized_pitch"], "optimized_roll": td["optimized_roll"],
        "U_yaw": fm["U_yaw"], "U_pitch": fm["U_pitch"], "U_roll": fm["U_roll"], "U_id": fm["U_id"],
    }
