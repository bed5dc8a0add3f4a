"""Oracle: per-frame post-processing of the video demo (TEST INFRASTRUCTURE).

Restates /root/reference/generatePose_on_video.py:
  rounding + EMA   :211,215-224   round(degrees, 2); s = 0.4*new + 0.6*s, first prediction seeds
  face centre/axes :73-124        centre = mean of landmarks 1/33/263 x frame size,
                                  jump gate 100 px, three axis end points (size 80)
Drawing (cv2.line/putText) is out of scope; the end points are what is drawn.
"""
from __future__ import annotations

import math
from math import cos, sin

import numpy as np

ALPHA = 0.4                 # :179
MAX_CENTER_JUMP = 100       # :136
AXIS_SIZE = 80              # :73 default


def ema_sequence(poses_rad: np.ndarray, alpha: float = ALPHA) -> np.ndarray:
    """poses_rad f32/f64[T,3] model outputs (radians) -> smoothed degrees f64[T,3]."""
    out = np.empty((len(poses_rad), 3), dtype=np.float64)
    s = None
    for t, pr in enumerate(poses_rad):
        new = [round(float(np.degrees(float(v))), 2) for v in pr]    # :211
        s = new if s is None else [alpha * n + (1 - alpha) * o for n, o in zip(new, s)]  # :215-222
        out[t] = s
    return out


def axes_on_face(prev_tdx, prev_tdy, frame_w, frame_h, nose, left_eye, right_eye, yaw, pitch, roll,
                 size=AXIS_SIZE, max_jump=MAX_CENTER_JUMP):
    """Returns (tdx, tdy, (x1,y1), (x2,y2), (x3,y3)) as floats; the reference draws int() of them."""
    pitch = pitch * np.pi / 180
    yaw = -(yaw * np.pi / 180)
    roll = roll * np.pi / 180
    # landmark coordinates are f32 values read as Python floats: the sums run in f64 (:91-92)
    new_tdx = (float(nose[0]) + float(left_eye[0]) + float(right_eye[0])) * frame_w / 3
    new_tdy = (float(nose[1]) + float(left_eye[1]) + float(right_eye[1])) * frame_h / 3
    if prev_tdx is None or prev_tdy is None:
        tdx, tdy = new_tdx, new_tdy
    else:
        dist = math.sqrt((new_tdx - prev_tdx) ** 2 + (new_tdy - prev_tdy) ** 2)
        tdx, tdy = (prev_tdx, prev_tdy) if dist > max_jump else (new_tdx, new_tdy)
    x1 = size * (cos(yaw) * cos(roll)) + tdx
    y1 = size * (cos(pitch) * sin(roll) + cos(roll) * sin(pitch) * sin(yaw)) + tdy
    x2 = size * (-cos(yaw) * sin(roll)) + tdx
    y2 = size * (cos(pitch) * cos(roll) - sin(pitch) * sin(yaw) * sin(roll)) + tdy
    x3 = size * (sin(yaw)) + tdx
    y3 = size * (-cos(yaw) * sin(pitch)) + tdy
    return tdx, tdy, (x1, y1), (x2, y2), (x3, y3)
