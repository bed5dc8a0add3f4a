#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the reference's own Python.

Run ONCE in the build container (needs /root/reference; the GPU box has neither
the reference nor a need to run this):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from /root/reference and called (file:line of the callee):
  FX1  helpers/FeatureExtractor.py:30   Read_Landmarks_and_Normalizing_using_IPD
  FX2  NLML_HPE_Model_Builder.py:71     AnglePredictionNetwork (+ shipped models/*.pth)
  FX2b NLML_HPE_Model_Builder.py:107    CombinedAnglePredictionModel on FX2's inputs (pass-through encoder)
  FX3  NLML_HPE_Model_Builder.py:26,107 LandmarkEncoder, CombinedAnglePredictionModel (eager + jit.script)
  FX3b the same at the reference's operating range (latent over the U_* row range, poses over the trained bins)
  FX3c FX3b's model on 16,384 faces: the reference's output as ONE batched call, in batches of 256 and one face at a time
  FX4  TD_Tester.py:31                  objective (and the same einsum for x_hat)
  FX5  TD_Tester.py:162                 Test (scipy Powell)
  FX6  NLML_HPE_Test.py:62,95           compute_maev, compute_errors
  FX7  generatePose_on_video.py:73,128  visualize_axes_on_face, process_video (EMA loop)
  FX8  NLML_HPE_MLPHeadsTrainer.py:71   cosine (called in the loops of :179-205 on the config_MlpHeads.yaml grids)
  FX9  TD_Tester.py:60                  compute_gradient (on the FX4 inputs)

cv2 / mediapipe are not installed; empty stub modules satisfy the imports, and
for FX7 the handful of cv2 / FaceMesh entry points process_video touches are
replaced by recording fakes so the reference's own frame loop runs on synthetic
landmarks.  Nothing of the reference's source is written to the fixtures: they
hold inputs, seeds and the numbers the reference returned.

Inputs come from nlml_hpe_amd.synth (Philox), so tests can regenerate them.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.environ.get("NLML_GOLDEN_OUT", HERE)      # where the fixtures are written (the regeneration test uses a temp dir)
# The repo root holds entry points with the reference's own file names (NLML_HPE_Model_Builder.py, NLML_HPE_Test.py,
# generatePose_on_video.py, TD_Inference.py), so the REFERENCE must come first on sys.path and the repo root must not
# shadow it: drop every path entry that resolves to the repo root (python puts the script dir / cwd there), then
# REF first, REPO after it (only for the packages the reference does not have: nlml_hpe_amd, oracle).
sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != REPO]
sys.path.insert(0, REF)
sys.path.insert(1, REPO)
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

for _n in ("cv2", "mediapipe", "utils", "tensorly"):
    sys.modules.setdefault(_n, types.ModuleType(_n))

import torch  # noqa: E402

from nlml_hpe_amd import synth  # noqa: E402

torch.set_num_threads(1)  # fixture values independent of the thread count of this box


def _ref(name: str):
    """Import a module of the reference and make sure it IS the reference's file (not a same-named file of this repo)."""
    import importlib
    mod = importlib.import_module(name)
    path = os.path.abspath(getattr(mod, "__file__", "") or "")
    if not path.startswith(REF + os.sep):
        raise ImportError(f"{name} resolved to {path}, not to the reference under {REF}")
    return mod


class _LM:
    """Duck-typed MediaPipe landmark: .x/.y/.z are Python floats of f32 values."""
    __slots__ = ("x", "y", "z")

    def __init__(self, p):
        self.x, self.y, self.z = float(p[0]), float(p[1]), float(p[2])


def fx1_normalise():
    FE = _ref("helpers.FeatureExtractor")
    lm = synth.raw_landmarks(16, seed=11)
    lm[3, 263] = lm[3, 33]                 # ipd == 0 -> 1e-6 branch (FeatureExtractor.py:47-48)
    lm[4] *= 1e-3                          # tiny face
    lm[5] = lm[5] * 1920.0                 # pixel-scale coordinates
    lm[6, 33] = lm[6, 263] + np.float32(1e-7)  # near-degenerate ipd
    out_norm = np.empty((16, 1404), np.float32)
    out_raw = np.empty((16, 1404), np.float32)
    for b in range(16):
        lms = [_LM(p) for p in lm[b]]
        ref = [lms[1].x, lms[1].y, lms[1].z]                              # FeatureExtractor.py:85-86
        l1 = FE.Read_Landmarks_and_Normalizing_using_IPD(lms, ref, True)
        l0 = FE.Read_Landmarks_and_Normalizing_using_IPD(lms, ref, False)
        out_norm[b] = torch.tensor(l1[0:1404]).float().numpy()            # :101
        out_raw[b] = torch.tensor(l0[0:1404]).float().numpy()
    np.savez_compressed(os.path.join(OUT, "fx1_normalise.npz"), landmarks=lm, features_norm=out_norm, features_raw=out_raw)
    print("FX1", out_norm.shape, float(np.abs(out_norm).max()))


def _load_heads(MB):
    heads = {}
    for n in ("yaw", "pitch", "roll"):
        net = MB.AnglePredictionNetwork(3)
        net.load_state_dict(torch.load(os.path.join(REF, "models", f"{n}_network.pth"), map_location="cpu"))
        heads[n] = net.eval()
    return heads


def fx2_heads():
    MB = _ref("NLML_HPE_Model_Builder")
    TD_Tester = _ref("TD_Tester")
    heads = _load_heads(MB)
    fm = np.load(os.path.join(REF, "outputs/features/Factor_Matrices.npz"))
    td = np.load(os.path.join(REF, "outputs/features/Trained_data.npz"))
    out = {}
    for n in ("yaw", "pitch", "roll"):
        U = fm[f"U_{n}"].astype(np.float32)
        P = td[f"optimized_{n}"]
        w = np.radians(np.linspace(-60, 60, 49))
        sweep = np.array([[TD_Tester.func(wi, p) for p in P] for wi in w]).astype(np.float32)   # cosine curves
        z = np.concatenate([U, sweep, np.zeros((1, 3), np.float32)], axis=0)
        with torch.no_grad():
            y = heads[n](torch.from_numpy(z)).numpy()
        out[f"in_{n}"] = z
        out[f"out_{n}"] = y
    np.savez_compressed(os.path.join(OUT, "fx2_heads.npz"), **out)
    print("FX2", {k: v.shape for k, v in out.items()})


def fx3_encoder_heads():
    MB = _ref("NLML_HPE_Model_Builder")
    heads = _load_heads(MB)
    out = {}
    for F in (1404, 136):
        enc = MB.LandmarkEncoder(F, [(1, 3)] * 3)
        sd = {k: torch.from_numpy(v) for k, v in synth.encoder_state_dict(F, seed=0).items()}
        enc.load_state_dict(sd)
        model = MB.CombinedAnglePredictionModel(enc, heads["yaw"], heads["pitch"], heads["roll"]).eval()
        x = synth.features(256, F, seed=1)
        x[7] = 0.0            # the all-zero "no face" row still flows through the reference's forward
        with torch.no_grad():
            y = torch.cat(model(torch.from_numpy(x)), dim=1).numpy()
            ys = torch.cat(torch.jit.script(model)(torch.from_numpy(x)), dim=1).numpy()
            y1 = torch.cat([torch.cat(model(torch.from_numpy(x[i:i + 1])), dim=1) for i in range(16)]).numpy()
        assert np.array_equal(y, ys), "eager != scripted"
        out[f"rad_F{F}"] = y
        out[f"rad_b1_F{F}"] = y1          # batch-1 calls, the way the reference runs (NLML_HPE_Test.py:262-272)
        out[f"x_crc_F{F}"] = np.array([float(x.astype(np.float64).sum()), float(np.abs(x).astype(np.float64).sum())])
        print("FX3", F, y.shape, np.degrees(np.abs(y).max()), "b1 vs batched max|d| deg", np.degrees(np.abs(y[:16] - y1).max()))
    np.savez_compressed(os.path.join(OUT, "fx3_encoder_heads.npz"), **out)


def fx3b_reference_range():
    """The encoder + heads where the reference actually operates: latent rows spanning the value range of the rows of
    U_yaw / U_pitch / U_roll (outputs/features/Factor_Matrices.npz, +-0.66) and poses across the trained bins
    (configs/config_EncoderTrainer.yaml:1-13: yaw +-50, pitch +-40, roll +-30 deg), 256 faces, F = 1404.

    Encoder: synth weights with hidden_weight_gain 2.0 (the face-to-face variation survives the stack), and the LAST
    layer (encoder.10) rescaled per output row (weight row * s_j, bias * s_j + t_j) so that latent column j spans
    [min U[:, j], max U[:, j]] over the 256 faces.  s, t are chosen here from the reference encoder's own f32 latent and
    the resulting encoder.10 tensors are stored in the fixture, so the tests rebuild exactly the weights the reference ran.
    Stored next to the batched outputs: the reference called one face at a time (how NLML_HPE_Test.py:262-272 runs it) and
    with 8 intra-op threads -- the reference's OWN spread between its call shapes, which bounds what "the reference's
    result" means at this range."""
    MB = _ref("NLML_HPE_Model_Builder")
    heads = _load_heads(MB)
    F, B = 1404, 256
    fm = np.load(os.path.join(REF, "outputs/features/Factor_Matrices.npz"))
    lo = np.concatenate([fm[f"U_{n}"].min(0) for n in ("yaw", "pitch", "roll")]).astype(np.float64)
    hi = np.concatenate([fm[f"U_{n}"].max(0) for n in ("yaw", "pitch", "roll")]).astype(np.float64)
    sd_np = synth.encoder_state_dict(F, seed=0, hidden_weight_gain=2.0)
    x = synth.features(B, F, seed=21)
    enc = MB.LandmarkEncoder(F, [(1, 3)] * 3)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    with torch.no_grad():
        lat = torch.cat([m.reshape(B, 3) for m in enc(torch.from_numpy(x))], dim=1).numpy().astype(np.float64)
    s = (hi - lo) / np.ptp(lat, axis=0)
    t = (hi + lo) / 2 - s * (lat.max(0) + lat.min(0)) / 2
    w10 = (sd_np["encoder.10.weight"].astype(np.float64) * s[:, None]).astype(np.float32)
    b10 = (sd_np["encoder.10.bias"].astype(np.float64) * s + t).astype(np.float32)
    sd_np["encoder.10.weight"], sd_np["encoder.10.bias"] = w10, b10
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model = MB.CombinedAnglePredictionModel(enc, heads["yaw"], heads["pitch"], heads["roll"]).eval()
    xt = torch.from_numpy(x)
    with torch.no_grad():
        y = torch.cat(model(xt), dim=1).numpy()
        lat2 = torch.cat([m.reshape(B, 3) for m in enc(xt)], dim=1).numpy()
        y1 = torch.cat([torch.cat(model(xt[i:i + 1]), dim=1) for i in range(B)]).numpy()
        torch.set_num_threads(8)
        y8 = torch.cat(model(xt), dim=1).numpy()
        torch.set_num_threads(1)
    np.savez_compressed(os.path.join(OUT, "fx3b_reference_range.npz"), rad=y, rad_b1=y1, rad_threads8=y8, latent=lat2,
                        enc10_weight=w10, enc10_bias=b10, u_lo=lo, u_hi=hi,
                        x_crc=np.array([float(x.astype(np.float64).sum()), float(np.abs(x).astype(np.float64).sum())]))
    print("FX3b latent span", lat2.min(0).round(3), lat2.max(0).round(3))
    print("FX3b pose deg span", np.degrees(y.min(0)).round(1), np.degrees(y.max(0)).round(1),
          "| reference vs itself: batch-1 %.2e deg, 8 threads %.2e deg"
          % (np.degrees(np.abs(y - y1).max()), np.degrees(np.abs(y - y8).max())))


def fx3c_reference_range_large():
    """FX3b's model (the reference's operating range) on 16,384 Philox faces -- enough faces for tail statistics
    (p50 / p99 / max of the distance from the f64 truth, fraction of faces beyond 1e-4 deg), which is how the GPU tests
    state parity there: "no worse than the reference itself".  Stored: the reference's poses from ONE batched call, from
    calls of 256 faces and from one-face calls (how NLML_HPE_Test.py:262-272 runs it); 1 intra-op thread.  The weights are
    FX3b's (encoder.10 read from that fixture), the inputs regenerate from the seed."""
    MB = _ref("NLML_HPE_Model_Builder")
    heads = _load_heads(MB)
    F, B = 1404, 16384
    src = os.path.join(OUT, "fx3b_reference_range.npz")
    g3b = np.load(src if os.path.exists(src) else os.path.join(HERE, "fx3b_reference_range.npz"))
    sd_np = synth.encoder_state_dict(F, seed=0, hidden_weight_gain=2.0)
    sd_np["encoder.10.weight"], sd_np["encoder.10.bias"] = g3b["enc10_weight"], g3b["enc10_bias"]
    enc = MB.LandmarkEncoder(F, [(1, 3)] * 3)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model = MB.CombinedAnglePredictionModel(enc, heads["yaw"], heads["pitch"], heads["roll"]).eval()
    x = synth.features(B, F, seed=23)
    xt = torch.from_numpy(x)
    with torch.no_grad():
        y = torch.cat(model(xt), dim=1).numpy()
        y256 = torch.cat([torch.cat(model(xt[i:i + 256]), dim=1) for i in range(0, B, 256)]).numpy()
        y1 = torch.cat([torch.cat(model(xt[i:i + 1]), dim=1) for i in range(B)]).numpy()
    np.savez_compressed(os.path.join(OUT, "fx3c_reference_range_16k.npz"), rad=y, rad_b256=y256, rad_b1=y1,
                        x_crc=np.array([float(x.astype(np.float64).sum()), float(np.abs(x).astype(np.float64).sum())]))
    d1 = np.degrees(np.abs(y - y1).max(1))
    print("FX3c pose deg span", np.degrees(y.min(0)).round(1), np.degrees(y.max(0)).round(1),
          "| reference vs itself (batched vs batch-1): max %.2e deg, faces > 1e-4 deg: %.4f" % (d1.max(), (d1 > 1e-4).mean()))


def fx2b_heads_through_model():
    """FX2's inputs (the heads' own operating points: rows of U_*, the trained cosine curves at -60..60 deg, zero) through the
    reference's WHOLE CombinedAnglePredictionModel, using the pass-through encoder of synth.passthrough_encoder_state_dict
    (latent_i = 2*tanh(x_i), x_i = atanh(z_i/2)); F = 136."""
    MB = _ref("NLML_HPE_Model_Builder")
    heads = _load_heads(MB)
    fx2 = np.load(os.path.join(HERE, "fx2_heads.npz"))
    zs = [fx2[f"in_{n}"] for n in ("yaw", "pitch", "roll")]
    B, F = max(len(z) for z in zs), 136
    z = np.concatenate([zz[np.arange(B) % len(zz)] for zz in zs], axis=1).astype(np.float64)      # [B,9]
    assert np.abs(z).max() < 1.9
    x = np.zeros((B, F), np.float32)
    x[:, :9] = np.arctanh(z / 2.0).astype(np.float32)
    x[:, 9:] = synth.features(B, F, seed=5)[:, 9:]            # the other inputs must not matter
    enc = MB.LandmarkEncoder(F, [(1, 3)] * 3)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.passthrough_encoder_state_dict(F, 2.0).items()})
    model = MB.CombinedAnglePredictionModel(enc, heads["yaw"], heads["pitch"], heads["roll"]).eval()
    with torch.no_grad():
        y = torch.cat(model(torch.from_numpy(x)), dim=1).numpy()
        lat = torch.cat([m.reshape(B, 3) for m in enc(torch.from_numpy(x))], dim=1).numpy()
    direct = np.concatenate([fx2[f"out_{n}"][np.arange(B) % len(fx2[f"in_{n}"])] for n in ("yaw", "pitch", "roll")], axis=1)
    np.savez_compressed(os.path.join(OUT, "fx2b_heads_through_model.npz"), x=x, z=z.astype(np.float32), latent=lat, rad=y)
    print("FX2b latent vs z max|d|", np.abs(lat - z).max(), "pose deg span", np.degrees(y.min(0)).round(1), np.degrees(y.max(0)).round(1),
          "vs the heads called directly (FX2) %.2e deg" % np.degrees(np.abs(y - direct).max()))


def _tucker_inputs(n, seed):
    from oracle import tucker as OT
    td = np.load(os.path.join(REF, "outputs/features/Trained_data.npz"))
    fm = np.load(os.path.join(REF, "outputs/features/Factor_Matrices.npz"))
    W = td["W"]
    idx = synth.tucker_grid_indices(n, seed=seed)
    g = synth.rng(seed, 77)
    X = np.stack([
        OT.grid_reconstruction(W, fm["U_id"][i], fm["U_yaw"][j], fm["U_pitch"][k], fm["U_roll"][l])
        for i, j, k, l in idx
    ])
    X = (X.astype(np.float64) + 1e-3 * g.standard_normal(X.shape)).astype(np.float32)
    return td, fm, W, idx, X


def fx4_td_objective():
    TD_Tester = _ref("TD_Tester")
    n = 32
    td, fm, W, idx, X = _tucker_inputs(n, seed=2)
    P = synth.tucker_params(n, 5, seed=2)
    P[0] = 0.0                                  # the optimiser's starting point (TD_Tester.py:166)
    Py, Pp, Pr = td["optimized_yaw"][0:3, :], td["optimized_pitch"][0:3, :], td["optimized_roll"][0:3, :]
    err = np.empty(n)
    xh = np.empty((8, 1404))
    for i in range(n):
        err[i] = TD_Tester.objective(P[i], W, torch.from_numpy(X[i]), Py, Pp, Pr)
        if i < 8:
            w_y, w_p, w_r = P[i][:3]
            f_y = np.array([TD_Tester.func(w_y, p) for p in Py]).flatten().astype(np.float32)
            f_p = np.array([TD_Tester.func(w_p, p) for p in Pp]).flatten().astype(np.float32)
            f_r = np.array([TD_Tester.func(w_r, p) for p in Pr]).flatten().astype(np.float32)
            xh[i] = np.einsum('ijklm,i,j,k,l->m', W, P[i][3:], f_y, f_p, f_r)      # TD_Tester.py:46
    np.savez_compressed(os.path.join(OUT, "fx4_td_objective.npz"), params=P, x=X, grid_idx=idx, err=err, x_hat=xh)
    print("FX4 err range", err.min(), err.max())


def fx5_td_end_to_end():
    TD_Tester = _ref("TD_Tester")
    import scipy
    td = np.load(os.path.join(REF, "outputs/features/Trained_data.npz"))
    fm = np.load(os.path.join(REF, "outputs/features/Factor_Matrices.npz"))
    from oracle import tucker as OT
    W = td["W"]
    Py, Pp, Pr = td["optimized_yaw"][0:3, :], td["optimized_pitch"][0:3, :], td["optimized_roll"][0:3, :]
    # grid poses (deg): yaw bins -50..50 step 10, pitch -40..40, roll -30..30 (configs/config_TD_main.yaml)
    picks = [(100, 7, 2, 4), (0, 5, 4, 3), (700, 0, 8, 6), (1500, 10, 0, 0)]   # (id, yaw-bin, pitch-bin, roll-bin)
    X, deg, nfev, fun, xs = [], [], [], [], []
    for (i, j, k, l) in picks:
        x = OT.grid_reconstruction(W, fm["U_id"][i], fm["U_yaw"][j], fm["U_pitch"][k], fm["U_roll"][l])
        del TD_Tester.objective_values[:]
        y, p, r, _ = TD_Tester.Test(W, torch.from_numpy(x), 5, Py, Pp, Pr, None, None, None, None)
        X.append(x); deg.append((y, p, r)); nfev.append(len(TD_Tester.objective_values))
        print("FX5", (i, j, k, l), (y, p, r), "nfev", nfev[-1])
    np.savez_compressed(os.path.join(OUT, "fx5_td_end_to_end.npz"), x=np.stack(X), picks=np.array(picks),
                        deg=np.array(deg), nfev=np.array(nfev), scipy_version=np.array(scipy.__version__))


def fx6_metrics():
    T = _ref("NLML_HPE_Test")
    gt = synth.poses_deg(64, seed=4)
    g = synth.rng(4, 99)
    pred = np.round(gt + 3.0 * g.standard_normal(gt.shape), 3)
    gt_l = [tuple(map(float, r)) for r in gt]
    pred_l = [tuple(map(float, r)) for r in pred]
    maev = T.compute_maev(gt_l, pred_l)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        T.compute_errors(gt_l, pred_l)
    small_gt = [(10.0, -5.0, 3.0), (-30.0, 20.0, -10.0)]
    small_pred = [(11.0, -4.0, 2.5), (-28.0, 22.0, -11.0)]
    small = T.compute_maev(small_gt, small_pred)
    R, l, b, f = T.W300_EulerAngles2Vectors(12.5, -33.0, 7.25)
    json.dump({
        "gt": gt.tolist(), "pred": pred.tolist(), "maev": list(map(float, maev)),
        "printed": buf.getvalue().splitlines(),
        "small_gt": small_gt, "small_pred": small_pred, "small_maev": list(map(float, small)),
        "euler_in": [12.5, -33.0, 7.25], "R": R.tolist(), "l": l.tolist(), "b": b.tolist(), "f": f.tolist(),
    }, open(os.path.join(OUT, "fx6_metrics.json"), "w"), indent=1)
    print("FX6", maev, small)


def fx7_video_math():
    """Run the reference's own process_video loop on synthetic landmarks with recording fakes."""
    import cv2
    import mediapipe as mp
    FE = _ref("helpers.FeatureExtractor")
    V = _ref("generatePose_on_video")

    T, Wpx, Hpx = 48, 1920, 1080
    g = synth.rng(7, 0)
    # smooth pose trajectory (radians) the fake model returns, one per frame
    t = np.linspace(0, 1, T)
    pose_rad = np.stack([0.6 * np.sin(2 * np.pi * t), 0.4 * np.cos(3 * np.pi * t), 0.3 * np.sin(5 * np.pi * t)], 1).astype(np.float32)
    lm = synth.raw_landmarks(T, seed=7) * 0.2 + 0.4          # faces near the frame centre
    lm[20, (1, 33, 263), :2] += 0.5                           # a >100 px jump -> gate keeps previous centre
    no_face = {5, 6}                                          # frames where FaceMesh "finds nothing"

    class _Res:
        def __init__(self, lms):
            self.multi_face_landmarks = lms

    class _Face:
        def __init__(self, arr):
            self.landmark = [_LM(p) for p in arr]

    state = {"frame": -1}

    class _FaceMesh:
        def __init__(self, *a, **k):
            pass

        def process(self, img):
            return _Res(None if state["frame"] in no_face else [_Face(lm[state["frame"]])])

    class _Cap:
        def __init__(self, src):
            self.i = 0

        def isOpened(self):
            return True

        def get(self, prop):
            return {0: 30.0, 1: Wpx, 2: Hpx}[prop]

        def read(self):
            if self.i >= T:
                return False, None
            state["frame"] = self.i
            self.i += 1
            return True, np.zeros((Hpx, Wpx, 3), np.uint8)

        def release(self):
            pass

    cv2.VideoCapture = _Cap
    cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT = 0, 1, 2
    cv2.COLOR_BGR2RGB = 0
    cv2.FONT_HERSHEY_SIMPLEX = 0
    cv2.cvtColor = lambda img, code: img
    cv2.line = lambda *a, **k: None
    cv2.putText = lambda *a, **k: None
    cv2.imshow = lambda *a, **k: None
    cv2.waitKey = lambda *a, **k: 0
    cv2.destroyAllWindows = lambda: None
    mp.solutions = types.SimpleNamespace(face_mesh=types.SimpleNamespace(FaceMesh=_FaceMesh))

    class _Model:
        def __call__(self, x):
            p = pose_rad[state["frame"]]
            return tuple(torch.tensor([[float(v)]]) for v in p)

    rec = []
    orig = V.visualize_axes_on_face

    def spy(prev_tdx, prev_tdy, max_jump, frame, landmarks, yaw, pitch, roll, size=80):
        # record what the reference's loop passes in (the EMA-smoothed angles) and what it gets back
        lines = []
        cv2.line = lambda fr, p0, p1, col, th: lines.append((p0, p1))
        out = orig(prev_tdx, prev_tdy, max_jump, frame, landmarks, yaw, pitch, roll, size)
        rec.append({"frame": state["frame"], "smoothed": [float(yaw), float(pitch), float(roll)],
                    "prev": [None if prev_tdx is None else float(prev_tdx), None if prev_tdy is None else float(prev_tdy)],
                    "centre": [float(out[1]), float(out[2])], "lines": [[list(map(int, a)), list(map(int, b))] for a, b in lines]})
        return out

    V.visualize_axes_on_face = spy
    with contextlib.redirect_stdout(io.StringIO()):
        V.process_video("synthetic.avi", None, _Model(), False, "cpu")
    V.visualize_axes_on_face = orig
    np.savez_compressed(os.path.join(OUT, "fx7_video_in.npz"), landmarks=lm, pose_rad=pose_rad, no_face=np.array(sorted(no_face)))
    json.dump({"width": Wpx, "height": Hpx, "frames": rec}, open(os.path.join(OUT, "fx7_video_math.json"), "w"))
    print("FX7 frames recorded", len(rec), "of", T)


def fx8_cosine_table():
    """The heads' training table: the reference's cosine() in the loops of NLML_HPE_MLPHeadsTrainer.py:179-205, on the
    angle grids of the reference's configs/config_MlpHeads.yaml and the shipped optimised cosine rows."""
    import yaml
    HT = _ref("NLML_HPE_MLPHeadsTrainer")
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "config_MlpHeads.yaml")))
    td = np.load(os.path.join(REPO, "outputs", "features", "Trained_data.npz"))
    out = {}
    for name in ("yaw", "pitch", "roll"):
        b = cfg[f"{name}_bins"]
        ang = np.radians(np.arange(b["min_bin"], b["max_bin"], b["interval"]).astype(np.float32))      # :179-181
        opt = td[f"optimized_{name}"]
        U = np.zeros((len(ang), opt.shape[0]))                                                        # :183-185
        for i, w in enumerate(ang):                                                                   # :189-205
            for j, row in enumerate(opt):
                a, b_, c, d = row
                U[i][j] = HT.cosine(w, a, b_, c, d)
        out[f"angles_{name}"] = ang
        out[f"U_{name}"] = U
    np.savez_compressed(os.path.join(OUT, "fx8_cosine_table.npz"), **out)
    print("FX8", {k: v.shape for k, v in out.items()})


def fx9_td_gradient():
    """The reference's analytic gradient on the FX4 inputs (same seeds => the test regenerates params and x from FX4)."""
    TD_Tester = _ref("TD_Tester")
    n = 32
    td, fm, W, idx, X = _tucker_inputs(n, seed=2)
    P = synth.tucker_params(n, 5, seed=2)
    P[0] = 0.0
    Py, Pp, Pr = td["optimized_yaw"][0:3, :], td["optimized_pitch"][0:3, :], td["optimized_roll"][0:3, :]
    G = np.stack([TD_Tester.compute_gradient(P[i], W, torch.from_numpy(X[i]), Py, Pp, Pr) for i in range(n)])
    np.savez_compressed(os.path.join(OUT, "fx9_td_gradient.npz"), grad=G)
    print("FX9 |grad| range", np.abs(G).min(), np.abs(G).max())


if __name__ == "__main__":
    which = sys.argv[1:] or ["1", "2", "2b", "3", "3b", "3c", "4", "5", "6", "7", "8", "9"]
    table = {"1": fx1_normalise, "2": fx2_heads, "2b": fx2b_heads_through_model, "3": fx3_encoder_heads,
             "3b": fx3b_reference_range, "3c": fx3c_reference_range_large, "4": fx4_td_objective,
             "5": fx5_td_end_to_end, "6": fx6_metrics, "7": fx7_video_math, "8": fx8_cosine_table, "9": fx9_td_gradient}
    for w in which:
        table[w]()
