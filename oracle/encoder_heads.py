"""Oracle: LandmarkEncoder + three AnglePredictionNetwork heads (TEST INFRASTRUCTURE).

Restates /root/reference/NLML_HPE_Model_Builder.py:
  LandmarkEncoder.forward            :55-68   Linear/ReLU x4, Linear/Tanh, Linear; latent split
  AnglePredictionNetwork.forward     :104-105 Linear/ReLU x4, Linear
  CombinedAnglePredictionModel.forward :115-126 encoder -> squeeze -> 3 heads -> radians

Three forms, all taking the reference's state-dict layout (weight [out,in]):
  forward_numpy(x, P, dtype)  plain ``h @ W.T + b`` in f32 (the parity oracle) or
                              f64 (the arithmetic truth both sides are compared to);
  forward_torch(x, P)         the same ATen ops the reference executes
                              (F.linear / relu / tanh) -- used as the timed CPU
                              baseline ("port") and as a second opinion;
  the C restatement in oracle/csrc/oracle.c (fixed k-ordered fmaf chain).

Outputs are radians, shape [B,3] = (yaw, pitch, roll) columns, i.e. the three
[B,1] tensors of the reference concatenated.
"""
from __future__ import annotations

import numpy as np

ENCODER_IDX = (0, 2, 4, 6, 8, 10)   # nn.Sequential slots holding Linear, Model_Builder.py:33-53
HEAD_IDX = (0, 2, 4, 6, 8)          # Model_Builder.py:76-92
HEAD_NAMES = ("yaw", "pitch", "roll")


class Params:
    """Encoder + heads weights as numpy f32 arrays in the reference's [out,in] layout."""

    def __init__(self, encoder_sd: dict, head_sds: dict, matrix_dims=((1, 3), (1, 3), (1, 3))):
        def _np(t):
            return np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
        self.enc = [(_np(encoder_sd[f"encoder.{i}.weight"]), _np(encoder_sd[f"encoder.{i}.bias"])) for i in ENCODER_IDX]
        self.heads = {
            n: [(_np(head_sds[n][f"model.{i}.weight"]), _np(head_sds[n][f"model.{i}.bias"])) for i in HEAD_IDX]
            for n in HEAD_NAMES
        }
        self.matrix_dims = tuple(matrix_dims)
        self.input_size = self.enc[0][0].shape[1]


def _lin(h, w, b, dtype):
    return h @ w.T.astype(dtype) + b.astype(dtype)


def encoder_latent_numpy(x: np.ndarray, p: Params, dtype=np.float32) -> np.ndarray:
    """[B,F] -> latent [B,9] (Model_Builder.py:58)."""
    h = np.asarray(x, dtype=dtype)
    n = len(p.enc)
    for li, (w, b) in enumerate(p.enc):
        h = _lin(h, w, b, dtype)
        if li < n - 2:
            h = np.maximum(h, 0)            # ReLU (:35,39,43,47)
        elif li == n - 2:
            h = np.tanh(h)                  # Tanh (:50)
    return h


def head_numpy(z: np.ndarray, layers, dtype=np.float32) -> np.ndarray:
    """[B,3] -> [B,1] (Model_Builder.py:76-92)."""
    h = np.asarray(z, dtype=dtype)
    for li, (w, b) in enumerate(layers):
        h = _lin(h, w, b, dtype)
        if li < len(layers) - 1:
            h = np.maximum(h, 0)
    return h


def forward_numpy(x: np.ndarray, p: Params, dtype=np.float32) -> np.ndarray:
    """[B,F] -> radians [B,3]; latent split as Model_Builder.py:60-66,118-124."""
    latent = encoder_latent_numpy(x, p, dtype)
    outs, start = [], 0
    for name, (m, n) in zip(HEAD_NAMES, p.matrix_dims):
        size = m * n
        z = latent[:, start:start + size]      # .view(-1,m,n).squeeze(1) with m == 1
        start += size
        outs.append(head_numpy(z, p.heads[name], dtype))
    return np.concatenate(outs, axis=1)


def forward_torch(x, p: Params, num_threads: int | None = None):
    """Same ATen op sequence as the reference, torch CPU f32.  Returns np f32 [B,3]."""
    import torch
    import torch.nn.functional as F
    if num_threads:
        torch.set_num_threads(int(num_threads))
    with torch.no_grad():
        h = torch.as_tensor(np.asarray(x, dtype=np.float32))
        enc = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in p.enc]
        n = len(enc)
        for li, (w, b) in enumerate(enc):
            h = F.linear(h, w, b)
            if li < n - 2:
                h = torch.relu(h)
            elif li == n - 2:
                h = torch.tanh(h)
        outs, start = [], 0
        for name, (m, nn_) in zip(HEAD_NAMES, p.matrix_dims):
            z = h[:, start:start + m * nn_]
            start += m * nn_
            layers = p.heads[name]
            for li, (w, b) in enumerate(layers):
                z = F.linear(z, torch.from_numpy(w), torch.from_numpy(b))
                if li < len(layers) - 1:
                    z = torch.relu(z)
            outs.append(z)
        return torch.cat(outs, dim=1).numpy()


def flops_per_face(input_size: int) -> int:
    """2 x MACs, bias/activation excluded."""
    enc = [input_size, 1024, 512, 256, 128, 64, 9]
    head = [3, 128, 256, 128, 64, 1]
    macs = sum(a * b for a, b in zip(enc[:-1], enc[1:])) + 3 * sum(a * b for a, b in zip(head[:-1], head[1:]))
    return 2 * macs


def _bf16_round(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even f32 -> bf16 -> f32 (what v_cvt_pk_bf16_f32 and the host packer do)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.astype(np.uint32).view(np.float32)


def forward_bf16_emulated(x: np.ndarray, p: Params) -> np.ndarray:
    """Model of the bf16 THROUGHPUT mode (encoder_heads_bf16_w8.hip): weights and every activation that goes
    through LDS rounded to bf16, products and sums in f64 (the kernel accumulates in f32), bias f32.
    Not a parity oracle for the reference -- it pins the throughput kernel's indexing and rounding points."""
    def lin(h, w, b):
        return h @ _bf16_round(w).T.astype(np.float64) + b.astype(np.float64)
    h = _bf16_round(np.asarray(x, dtype=np.float32)).astype(np.float64)
    n = len(p.enc)
    for li, (w, b) in enumerate(p.enc):
        h = lin(h, w, b)
        if li < n - 2:
            h = np.maximum(h, 0)
        elif li == n - 2:
            h = np.tanh(h)
        if li < n - 1:
            h = _bf16_round(h.astype(np.float32)).astype(np.float64)
    latent = h
    lat_b = _bf16_round(latent.astype(np.float32)).astype(np.float64)
    outs, start = [], 0
    for name, (m, nn_) in zip(HEAD_NAMES, p.matrix_dims):
        z = lat_b[:, start:start + m * nn_]
        start += m * nn_
        layers = p.heads[name]
        for li, (w, b) in enumerate(layers):
            z = lin(z, w, b)
            if li < len(layers) - 1:
                z = _bf16_round(np.maximum(z, 0).astype(np.float32)).astype(np.float64)
        outs.append(z)
    return np.concatenate(outs, axis=1), latent
