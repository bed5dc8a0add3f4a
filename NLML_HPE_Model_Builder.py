#!/usr/bin/env python3
"""Compose the encoder and the three head networks into the deployable model -- MI355X counterpart of
the reference's NLML_HPE_Model_Builder.py (model_builder(), :168): reads configs/config_EncoderTrainer.yaml
(input_size), outputs/features/*.npz (head input size = optimized_*.shape[0], latent split =
U_*.shape[1], :178-195) and the four state dicts (:201-216), validates them against the kernel's
architecture and writes the packed MFMA-fragment blob next to where the reference writes its scripted file.

    python NLML_HPE_Model_Builder.py [--synthetic-encoder-seed 0]
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from nlml_hpe_amd import synth, weights
from nlml_hpe_amd.entrypoints import load_config


def model_builder(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic-encoder-seed", type=int, default=None,
                    help="use generator-G encoder weights when models/Encoder.pth is absent (the reference ships none)")
    ap.add_argument("--out", default="models/combined_model_packed.nlml")
    ap.add_argument("--mode", choices=["f16x2s", "f32", "f16x2", "bf16"], default="f16x2s",
                    help="kernel mode the blob is packed for (the forward entry points recognise it by the blob size)")
    args = ap.parse_args(argv)

    input_size = load_config("configs/config_EncoderTrainer.yaml")["input_size"]
    art = weights.load_tucker_artefacts("outputs/features")
    head_in = [art[f"optimized_{n}"].shape[0] for n in ("yaw", "pitch", "roll")]
    dims = [(1, art[f"U_{n}"].shape[1]) for n in ("yaw", "pitch", "roll")]
    if head_in != [3, 3, 3] or dims != [(1, 3)] * 3:
        raise SystemExit(f"artefacts describe head inputs {head_in} / latent split {dims}; the kernel is built for 3 x (1,3)")
    heads = weights.load_head_state_dicts("models")
    try:
        enc = weights.load_encoder_state_dict("models")
    except FileNotFoundError:
        if args.synthetic_encoder_seed is None:
            raise
        enc = synth.encoder_state_dict(input_size, args.synthetic_encoder_seed)
    from nlml_hpe_amd import _lib
    blob = weights.pack_blob(enc, heads, _lib.mode_from_name(args.mode))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    np.asarray(blob).tofile(args.out)
    print(f"model is built: {args.out} ({blob.nbytes} bytes, F={input_size}, mode={args.mode})")
    return args.out


if __name__ == "__main__":
    model_builder()
