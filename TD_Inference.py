#!/usr/bin/env python3
"""Head pose by the Tucker-decomposition path -- MI355X counterpart of the reference's TD_Inference.py
(inference(), :20): loads W and the cosine parameters from outputs/features/*.npz exactly as :40-51 and
runs TD_Tester.Test; the 2-4 s/face scipy Powell loop runs as one device-side lock-step launch.

    python TD_Inference.py --image_path faces.npy     # pre-extracted FaceMesh landmarks f32[N,468,3]
(the flag keeps the reference's name; image files need MediaPipe, which this image does not have).
"""
from __future__ import annotations

import argparse
import time

import torch

from nlml_hpe_amd import TD_Tester, ops, weights
from nlml_hpe_amd.entrypoints import load_landmarks


def inference(argv=None):
    ap = argparse.ArgumentParser(description="Inference the head pose of input faces by the TD path")
    ap.add_argument("--image_path", type=str, required=True, help="landmarks .npy/.npz (f32[N,468,3])")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--order", choices=["reference", "fast"], default="reference",
                    help="objective's operation order: reference = the reference's bits and scipy's own end point (default); "
                         "fast = f64 matrix cores (~3x the faces/s; NOT a parity mode: the end point is sensitive to the last bits of the "
                         "objective -- 6e-3 deg from scipy on clean grid faces, up to degrees on noisy ones)")
    args = ap.parse_args(argv)
    torch.cuda.set_device(torch.device(args.device))

    raw = load_landmarks(args.image_path)
    x = ops.normalize_ipd(torch.from_numpy(raw).cuda(), True)                  # get_feature_vector(..., normalize=True), :37
    art = weights.load_tucker_artefacts("./outputs/features")
    u_id_shape = art["U_id"][1].size                                            # :51
    t0 = time.time()
    deg = TD_Tester.Test_batch(art["W"], x, u_id_shape, art["optimized_yaw"][0:3, :], art["optimized_pitch"][0:3, :],
                               art["optimized_roll"][0:3, :], order=args.order)  # :56
    dt = time.time() - t0
    for i, (y, p, r) in enumerate(deg):
        tag = f"[{i}] " if len(deg) > 1 else ""
        print(f"{tag}Estimated yaw in degree = {y:.2f}")
        print(f"{tag}Estimated pitch in degree = {p:.2f}")
        print(f"{tag}Estimated roll in degree = {r:.2f}")
    print(f"({len(deg)} faces in {dt:.3f} s)")
    return deg


if __name__ == "__main__":
    inference()
