"""Generator G: build-owned, counter-based synthetic inputs and encoder weights.

The reference ships no ``models/Encoder.pth`` and no landmark
data, so every benchmark / parity input is synthesised here from numpy's
Philox bit generator.  Philox is counter based: the same (seed, stream) gives
bit-identical arrays on any box with the same numpy, which is what lets the
golden fixtures generated in the build container be regenerated on the GPU box
(SURVEY.md 8d "Generator G").

Shapes follow the reference:
  * raw landmarks  f32[B,468,3]  -- MediaPipe FaceMesh normalised coordinates
    (helpers/FeatureExtractor.py:30-66 reads .x/.y/.z of 468 landmarks)
  * features       f32[B,F]      -- IPD-normalised, flattened x,y,z
    (helpers/FeatureExtractor.py:101)
  * encoder layers Linear(F,1024) .. Linear(64,9)
    (NLML_HPE_Model_Builder.py:33-53), PyTorch default init
    kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)).
"""
from __future__ import annotations

import numpy as np

NUM_LANDMARKS = 468
F_REFERENCE = NUM_LANDMARKS * 3  # 1404, configs/config_EncoderTrainer.yaml input_size
ENCODER_WIDTHS = (1024, 512, 256, 128, 64, 9)  # NLML_HPE_Model_Builder.py:33-53

# stream ids (Philox key = [seed, stream])
_STREAM_ENCODER = 0
_STREAM_LANDMARKS = 1
_STREAM_FEATURES = 2
_STREAM_TUCKER = 3
_STREAM_POSE = 4


def rng(seed: int, stream: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def _uniform_f32(g: np.random.Generator, shape, lo: float, hi: float) -> np.ndarray:
    # random() yields float64 in [0,1) from the raw Philox words; the affine map
    # and the single rounding to f32 are IEEE operations => portable bits.
    return (lo + (hi - lo) * g.random(shape)).astype(np.float32)


def encoder_state_dict(input_size: int = F_REFERENCE, seed: int = 0, hidden_weight_gain: float = 1.0) -> dict:
    """Synthetic encoder weights under the reference's state-dict keys.

    Keys ``encoder.{0,2,4,6,8,10}.{weight,bias}``, weight [out,in] f32 -- the
    layout ``LandmarkEncoder.load_state_dict`` expects
    (NLML_HPE_Model_Builder.py:33-53,202).

    hidden_weight_gain multiplies the WEIGHTS (not the biases) of the five hidden layers.  PyTorch's default init
    shrinks the face-to-face variation by ~0.4 per ReLU layer, so the default encoder's latent barely depends on the
    face (std 0.007); gain 2.0 keeps the variation alive through the stack (a conditioning closer to a trained
    encoder's), which is what the reference-range fixture FX3b uses.  The gain is a power of two, so it is exact.
    """
    g = rng(seed, _STREAM_ENCODER)
    sd = {}
    fan_in = int(input_size)
    for i, width in enumerate(ENCODER_WIDTHS):
        bound = 1.0 / np.sqrt(fan_in)
        w = _uniform_f32(g, (width, fan_in), -bound, bound)
        if i < len(ENCODER_WIDTHS) - 1 and hidden_weight_gain != 1.0:
            w = w * np.float32(hidden_weight_gain)
        sd[f"encoder.{2 * i}.weight"] = w
        sd[f"encoder.{2 * i}.bias"] = _uniform_f32(g, (width,), -bound, bound)
        fan_in = width
    return sd


def passthrough_encoder_state_dict(input_size: int = 136, out_gain: float = 2.0) -> dict:
    """An encoder that hands its first nine inputs through to the latent: latent_i = out_gain * tanh(x_i).

    Layer 0 forms relu(x_i) and relu(-x_i) (i < 9), the hidden layers carry those 18 non-negative values unchanged,
    the Tanh layer recombines them into x_i and the last layer multiplies by out_gain.  With x_i = atanh(z_i / out_gain)
    the three heads see z -- which is how fixture FX2b puts the heads' own operating points (rows of U_yaw/U_pitch/U_roll
    and the trained cosine curves, tests/golden/fx2_heads.npz) through the whole fused forward on the GPU.
    """
    sd = {}
    fan_in = int(input_size)
    if fan_in < 9:
        raise ValueError("passthrough encoder needs at least 9 inputs")
    for i, width in enumerate(ENCODER_WIDTHS):
        w = np.zeros((width, fan_in), np.float32)
        if i == 0:
            for k in range(9):
                w[k, k], w[9 + k, k] = 1.0, -1.0
        elif i < 4:
            for k in range(18):
                w[k, k] = 1.0
        elif i == 4:
            for k in range(9):
                w[k, k], w[k, 9 + k] = 1.0, -1.0
        else:
            for k in range(9):
                w[k, k] = out_gain
        sd[f"encoder.{2 * i}.weight"] = w
        sd[f"encoder.{2 * i}.bias"] = np.zeros((width,), np.float32)
        fan_in = width
    return sd


def raw_landmarks(batch: int, seed: int = 1) -> np.ndarray:
    """f32[B,468,3] raw FaceMesh-like coordinates, U(0,1)."""
    return _uniform_f32(rng(seed, _STREAM_LANDMARKS), (batch, NUM_LANDMARKS, 3), 0.0, 1.0)


def features(batch: int, input_size: int = F_REFERENCE, seed: int = 1) -> np.ndarray:
    """f32[B,F] IPD-normalised-looking features: U(-2,2), landmark 1 == origin.

    After IPD normalisation the nose tip (landmark 1, columns 3:6) is exactly
    zero (helpers/FeatureExtractor.py:55-57); keep that structure when F>=6.
    """
    x = _uniform_f32(rng(seed, _STREAM_FEATURES), (batch, input_size), -2.0, 2.0)
    if input_size >= 6:
        x[:, 3:6] = 0.0
    return x


def tucker_params(n: int, u_id_dim: int = 5, seed: int = 2) -> np.ndarray:
    """f64[N, 3+u_id_dim] objective parameters.

    (w_y, w_p, w_r) ~ U(+-0.9, +-0.7, +-0.5) rad, u_id ~ N(0, 0.05).
    """
    g = rng(seed, _STREAM_TUCKER)
    p = np.empty((n, 3 + u_id_dim), dtype=np.float64)
    span = np.array([0.9, 0.7, 0.5])
    p[:, :3] = (2.0 * g.random((n, 3)) - 1.0) * span
    p[:, 3:] = 0.05 * g.standard_normal((n, u_id_dim))
    return p


def tucker_grid_indices(n: int, shape=(1620, 11, 9, 7), seed: int = 2) -> np.ndarray:
    """int64[N,4] (identity, yaw-bin, pitch-bin, roll-bin) picks for grid reconstructions."""
    g = rng(seed, _STREAM_TUCKER + 100)
    return np.stack([g.integers(0, s, size=n) for s in shape], axis=1)


def tucker_grid_faces(art: dict, idx: np.ndarray, noise: float = 1e-3, seed: int = 2) -> np.ndarray:
    """f32[N,1404] training-grid faces W x1 U_id[i] x2 U_yaw[j] x3 U_pitch[k] x4 U_roll[l] plus N(0, noise) --
    the synthetic inputs of BASELINE.json config 3."""
    W = np.asarray(art["W"], np.float64).reshape(5, 3, 3, 3, -1)
    i, j, k, l = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
    c = np.einsum("na,nb,nc,nd->nabcd", np.asarray(art["U_id"], np.float64)[i], np.asarray(art["U_yaw"], np.float64)[j],
                  np.asarray(art["U_pitch"], np.float64)[k], np.asarray(art["U_roll"], np.float64)[l])
    X = (c.reshape(len(idx), -1) @ W.reshape(135, -1)).astype(np.float32)
    return (X.astype(np.float64) + noise * rng(seed, 77).standard_normal(X.shape)).astype(np.float32)


def poses_deg(n: int, seed: int = 4) -> np.ndarray:
    """f64[N,3] smooth-ish ground-truth poses in degrees inside the reference's bin ranges."""
    g = rng(seed, _STREAM_POSE)
    span = np.array([50.0, 40.0, 30.0])
    return (2.0 * g.random((n, 3)) - 1.0) * span
