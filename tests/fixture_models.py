"""Rebuild the inputs and weights of the reference-range fixtures (FX3b, FX2b) -- test helper.

The fixtures hold the reference's outputs plus the few tensors that cannot be regenerated from a seed
(tests/golden/make_golden.py: fx3b_reference_range, fx2b_heads_through_model)."""
import os

import numpy as np

from nlml_hpe_amd import synth


def fx3b(golden_dir):
    """-> (g, encoder state dict, x f32[256,1404]); g["rad"] = the reference's batched output on exactly these."""
    g = np.load(os.path.join(golden_dir, "fx3b_reference_range.npz"))
    sd = synth.encoder_state_dict(1404, seed=0, hidden_weight_gain=2.0)
    sd["encoder.10.weight"], sd["encoder.10.bias"] = g["enc10_weight"], g["enc10_bias"]
    x = synth.features(256, 1404, seed=21)
    assert float(x.astype(np.float64).sum()) == g["x_crc"][0]
    return g, sd, x


def fx3c(golden_dir):
    """-> (g, encoder state dict, x f32[16384,1404]): FX3b's weights on 16,384 faces; g["rad"] / g["rad_b256"] / g["rad_b1"] = the
    reference's poses from one batched call, from calls of 256 faces and from one-face calls."""
    g = np.load(os.path.join(golden_dir, "fx3c_reference_range_16k.npz"))
    _, sd, _ = fx3b(golden_dir)
    x = synth.features(16384, 1404, seed=23)
    assert float(x.astype(np.float64).sum()) == g["x_crc"][0]
    return g, sd, x


def error_stats(got, truth):
    """Per-face max over (yaw, pitch, roll) of |got - truth| in degrees -> dict(p50, p99, max, frac_above_1e-4)."""
    d = np.degrees(np.abs(np.asarray(got, np.float64) - np.asarray(truth, np.float64))).max(axis=1)
    return {"p50": float(np.percentile(d, 50)), "p99": float(np.percentile(d, 99)), "max": float(d.max()),
            "frac_above_1e-4": float((d > 1e-4).mean())}


def fx2b(golden_dir):
    """-> (g, encoder state dict, x f32[61,136]); the pass-through encoder puts FX2's head inputs on the latent."""
    g = np.load(os.path.join(golden_dir, "fx2b_heads_through_model.npz"))
    return g, synth.passthrough_encoder_state_dict(136, 2.0), g["x"]
