"""Oracle: evaluation metrics (TEST INFRASTRUCTURE).

Restates /root/reference/NLML_HPE_Test.py:
  W300_EulerAngles2Vectors :28-58   R = Rx(pitch) @ Ry(-yaw) @ Rz(roll), columns l/b/f
  compute_maev             :62-93   per-vector acos(clip(dot)) in degrees, means
  compute_errors           :95-130  MAE per axis, std(ddof=1) of abs errors, total MAE
"""
from __future__ import annotations

import math
import numpy as np


def euler_to_vectors(rx, ry, rz):
    rx, ry, rz = np.radians(rx), np.radians(ry), np.radians(rz)
    ry = -ry                                                          # :41
    R_x = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(rx), -np.sin(rx)], [0.0, np.sin(rx), np.cos(rx)]])
    R_y = np.array([[np.cos(ry), 0.0, np.sin(ry)], [0.0, 1.0, 0.0], [-np.sin(ry), 0.0, np.cos(ry)]])
    R_z = np.array([[np.cos(rz), -np.sin(rz), 0.0], [np.sin(rz), np.cos(rz), 0.0], [0.0, 0.0, 1.0]])
    R = R_x @ R_y @ R_z                                               # :54
    return R, R @ np.array([1, 0, 0]).T, R @ np.array([0, 1, 0]).T, R @ np.array([0, 0, 1]).T


def compute_maev(ground_truth, predicted):
    s1 = s2 = s3 = 0.0
    count = len(ground_truth)
    c = 180.0 / np.pi
    for i in range(count):
        yg, pg, rg = ground_truth[i]
        yp, pp, rp = predicted[i]
        _, lg, bg, fg = euler_to_vectors(pg, yg, rg)                  # (pitch, yaw, roll) order, :75-76
        _, lp, bp, fp = euler_to_vectors(pp, yp, rp)
        s1 += math.acos(np.clip(np.sum(lg * lp), -1, 1)) * c
        s2 += math.acos(np.clip(np.sum(bg * bp), -1, 1)) * c
        s3 += math.acos(np.clip(np.sum(fg * fp), -1, 1)) * c
    return (s1 + s2 + s3) / (3 * count), s1 / count, s2 / count, s3 / count


def compute_errors(true_angles, pred_angles) -> dict:
    """Same numbers the reference prints at :119-129, returned as a dict."""
    t = np.asarray(true_angles, dtype=np.float64)
    p = np.asarray(pred_angles, dtype=np.float64)
    e = np.abs(t - p)
    mae = e.mean(axis=0)
    std = e.std(axis=0, ddof=1)                                       # :108-110
    maev, l, d, f = compute_maev(true_angles, pred_angles)
    return {
        "mae_yaw": mae[0], "mae_pitch": mae[1], "mae_roll": mae[2], "mae_total": mae.sum() / 3,
        "maev": maev, "v_left": l, "v_down": d, "v_front": f,
        "std_yaw": std[0], "std_pitch": std[1], "std_roll": std[2],
    }


def interval_mae(true_angles, pred_angles, axis: int, intervals) -> dict:
    """MAE per [low, high) ground-truth interval (:143-152), without the plotting."""
    t = np.asarray(true_angles, dtype=np.float64)
    p = np.asarray(pred_angles, dtype=np.float64)
    out = {}
    for low, high in intervals:
        m = (t[:, axis] >= low) & (t[:, axis] < high)
        if m.any():
            out[(low, high)] = float(np.mean(np.abs(t[m, axis] - p[m, axis])))
    return out
