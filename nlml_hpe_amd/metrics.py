"""Evaluation metrics of the test entry point, vectorised on the device.

Same quantities as the reference's per-sample Python loops (NLML_HPE_Test.py):
  euler_to_vectors   W300_EulerAngles2Vectors :28-58   R = Rx(pitch) @ Ry(-yaw) @ Rz(roll)
  compute_maev       :62-93    mean angle between the left/down/front vectors of GT and prediction
  compute_errors     :95-130   per-axis MAE, std(ddof=1) of absolute errors, total MAE, MAEV
  interval_mae       :143-152  MAE per ground-truth interval (the numbers behind the plots)
All in float64 torch ops on whatever device the inputs live on (N x 3 tensors of degrees, columns
yaw, pitch, roll).  Plotting (matplotlib) is out of scope.
"""
from __future__ import annotations

import math

import torch


def _as64(a, device=None) -> torch.Tensor:
    t = torch.as_tensor(a, dtype=torch.float64)
    return t.to(device) if device is not None else t


def euler_to_vectors(pitch_deg: torch.Tensor, yaw_deg: torch.Tensor, roll_deg: torch.Tensor):
    """Batched W300_EulerAngles2Vectors(rx=pitch, ry=yaw, rz=roll) -> (R[N,3,3], l[N,3], b[N,3], f[N,3])."""
    rx, ry, rz = torch.deg2rad(pitch_deg), -torch.deg2rad(yaw_deg), torch.deg2rad(roll_deg)
    one, zero = torch.ones_like(rx), torch.zeros_like(rx)
    Rx = torch.stack([one, zero, zero, zero, rx.cos(), -rx.sin(), zero, rx.sin(), rx.cos()], -1).reshape(-1, 3, 3)
    Ry = torch.stack([ry.cos(), zero, ry.sin(), zero, one, zero, -ry.sin(), zero, ry.cos()], -1).reshape(-1, 3, 3)
    Rz = torch.stack([rz.cos(), -rz.sin(), zero, rz.sin(), rz.cos(), zero, zero, zero, one], -1).reshape(-1, 3, 3)
    R = Rx @ Ry @ Rz
    return R, R[:, :, 0], R[:, :, 1], R[:, :, 2]


def compute_maev(ground_truth, predicted):
    """-> (MAEV, left, down, front) as Python floats, like the reference's return tuple (:93)."""
    gt, pr = _as64(ground_truth), _as64(predicted)
    pr = pr.to(gt.device)
    _, lg, bg, fg = euler_to_vectors(gt[:, 1], gt[:, 0], gt[:, 2])
    _, lp, bp, fp = euler_to_vectors(pr[:, 1], pr[:, 0], pr[:, 2])
    c = 180.0 / math.pi
    e1 = torch.acos(torch.clamp((lg * lp).sum(1), -1, 1)) * c
    e2 = torch.acos(torch.clamp((bg * bp).sum(1), -1, 1)) * c
    e3 = torch.acos(torch.clamp((fg * fp).sum(1), -1, 1)) * c
    n = gt.shape[0]
    s1, s2, s3 = e1.sum().item(), e2.sum().item(), e3.sum().item()
    return (s1 + s2 + s3) / (3 * n), s1 / n, s2 / n, s3 / n


def compute_errors(true_angles, pred_angles, verbose: bool = True) -> dict:
    t, p = _as64(true_angles), _as64(pred_angles)
    p = p.to(t.device)
    e = (t - p).abs()
    mae = e.mean(0)
    std = e.std(0, unbiased=True)                                  # np.std(ddof=1), :108-110
    maev, l, d, f = compute_maev(t, p)
    out = {"mae_yaw": mae[0].item(), "mae_pitch": mae[1].item(), "mae_roll": mae[2].item(),
           "mae_total": (mae.sum() / 3).item(), "maev": maev, "v_left": l, "v_down": d, "v_front": f,
           "std_yaw": std[0].item(), "std_pitch": std[1].item(), "std_roll": std[2].item()}
    if verbose:                                                    # the reference's printed block, :119-129
        print(f'MAE (Yaw): {out["mae_yaw"]:.2f}')
        print(f'MAE (Pitch): {out["mae_pitch"]:.2f}')
        print(f'MAE (Roll): {out["mae_roll"]:.2f}')
        print(f'Total MAE: {out["mae_total"]:.2f}')
        print(f'MAEV: {out["maev"]:.2f}')
        print(f'Left vector Error (red): {out["v_left"]:.2f}')
        print(f'Down vector Error (green): {out["v_down"]:.2f}')
        print(f'Front vector Error (blue): {out["v_front"]:.2f}')
        print(f'std (Yaw): {out["std_yaw"]:.2f}')
        print(f'std (Pitch): {out["std_pitch"]:.2f}')
        print(f'std (Roll): {out["std_roll"]:.2f}')
    return out


def compute_interval_mae(true_angles, pred_angles, yaw_intervals, pitch_intervals, roll_intervals, label="NLML_HPE") -> dict:
    """Keys follow the reference's results dict (:148): '<Name> (<low>, <high>) - <label>'."""
    t, p = _as64(true_angles), _as64(pred_angles)
    p = p.to(t.device)
    res = {}
    for name, idx, intervals in (("Yaw", 0, yaw_intervals), ("Pitch", 1, pitch_intervals), ("Roll", 2, roll_intervals)):
        for low, high in intervals:
            m = (t[:, idx] >= low) & (t[:, idx] < high)
            if bool(m.any()):
                res[f"{name} ({low}, {high}) - {label}"] = (t[m, idx] - p[m, idx]).abs().mean().item()
    return res
