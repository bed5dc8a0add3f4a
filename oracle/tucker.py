"""Oracle: Tucker-einsum objective and its Powell driver (TEST INFRASTRUCTURE).

Restates /root/reference/TD_Tester.py:
  func        :25-28   a*cos(b*w+c)+d                       (f64)
  objective   :31-58   f-vectors (f64 -> f32), x_hat = einsum('ijklm,i,j,k,l->m'),
                       0.5*sum((x-x_hat)**2)                (einsum promotes to f64)
  Test        :162-199 scipy.optimize.minimize(method='Powell') from zeros(3+u), degrees

The debug-list appends (:52-56) and the unused ``jac=`` (:194, ignored by Powell)
are not part of the arithmetic and are dropped.
"""
from __future__ import annotations

import numpy as np


def func(w, params):
    a, b, c, d = params
    return a * np.cos(b * w + c) + d                               # :25-28


def f_vectors(params, params_y, params_p, params_r):
    """(f_y, f_p, f_r) as f32 arrays, exactly as :36-43."""
    w_y, w_p, w_r = params[:3]
    f_y = np.array([func(w_y, p) for p in params_y]).flatten().astype(np.float32)
    f_p = np.array([func(w_p, p) for p in params_p]).flatten().astype(np.float32)
    f_r = np.array([func(w_r, p) for p in params_r]).flatten().astype(np.float32)
    return f_y, f_p, f_r


def x_hat(params, W, params_y, params_p, params_r):
    """f64[1404] reconstruction, the same einsum call as :46."""
    u_id = params[3:]
    f_y, f_p, f_r = f_vectors(params, params_y, params_p, params_r)
    return np.einsum('ijklm,i,j,k,l->m', W, u_id, f_y, f_p, f_r)


def objective(params, W, x, params_y, params_p, params_r):
    """f64 scalar, :31-58.  ``x`` is a numpy f32 vector (the reference calls x.numpy())."""
    xh = x_hat(params, W, params_y, params_p, params_r)
    return 0.5 * np.sum((np.asarray(x) - xh) ** 2)                  # :49


def coefficients(params, params_y, params_p, params_r):
    """f64[I*J*K*L] outer product c = u (x) f_y (x) f_p (x) f_r, row-major (i,j,k,l).

    x_hat == c @ W.reshape(-1, M): the GEMV/GEMM form the HIP kernel uses
    (summation order differs from einsum's => agreement to ~1e-15 rel., not bits).
    """
    u_id = np.asarray(params[3:], dtype=np.float64)
    f_y, f_p, f_r = f_vectors(params, params_y, params_p, params_r)
    c = np.einsum('i,j,k,l->ijkl', u_id, f_y.astype(np.float64), f_p.astype(np.float64), f_r.astype(np.float64))
    return c.reshape(-1)


def objective_batch(P, W, X, params_y, params_p, params_r, return_xhat=False):
    """Vectorised GEMM form over N faces: P f64[N,3+u], X f32[N,M] -> err f64[N]."""
    P = np.asarray(P, dtype=np.float64)
    C = np.stack([coefficients(p, params_y, params_p, params_r) for p in P])
    Wm = np.asarray(W).reshape(-1, np.asarray(W).shape[-1]).astype(np.float64)
    XH = C @ Wm
    err = 0.5 * np.sum((np.asarray(X, dtype=np.float64) - XH) ** 2, axis=1)
    return (err, XH) if return_xhat else err


def grid_reconstruction(W, U_id_row, U_yaw_row, U_pitch_row, U_roll_row):
    """x = W x1 u_id x2 u_yaw x3 u_pitch x4 u_roll -> f32[1404] (a training-grid face)."""
    return np.einsum('ijklm,i,j,k,l->m', W, U_id_row, U_yaw_row, U_pitch_row, U_roll_row).astype(np.float32)


def test_powell(W, x, u_id_shape, params_y, params_p, params_r, return_result=False):
    """``Test`` :162-199: Powell from zeros, angles in degrees."""
    from scipy.optimize import minimize
    x0 = np.zeros(3 + u_id_shape)                                   # :166
    res = minimize(objective, x0, args=(W, np.asarray(x), params_y, params_p, params_r), method='Powell')  # :191-194
    deg = np.degrees(res.x)                                         # :196
    out = (deg[0], deg[1], deg[2])
    return (out, res) if return_result else out


def compute_gradient(params, W, x, params_y, params_p, params_r):
    """``compute_gradient`` :60-102, the analytic gradient handed to ``minimize(jac=...)`` at :194 (Powell ignores it).

    Restated as written, including the identity-mode term exactly as the reference forms it (:96): the inner einsum
    'ijklm,j,k,l->m' also sums over i."""
    params = np.asarray(params, dtype=np.float64)
    w_y, w_p, w_r = params[:3]
    u_id = params[3:]
    f_y, f_p, f_r = f_vectors(params, params_y, params_p, params_r)
    xh = np.einsum('ijklm,i,j,k,l->m', W, u_id, f_y, f_p, f_r)
    residuals = np.asarray(x) - xh

    def dfun(w, rows):
        return np.array([-p[0] * p[1] * np.sin(p[1] * w + p[2]) for p in rows]).flatten().astype(np.float32)

    g_y = -np.sum(residuals * np.einsum('ijklm,i,j,k,l->m', W, u_id, dfun(w_y, params_y), f_p, f_r))
    g_p = -np.sum(residuals * np.einsum('ijklm,i,j,k,l->m', W, u_id, f_y, dfun(w_p, params_p), f_r))
    g_r = -np.sum(residuals * np.einsum('ijklm,i,j,k,l->m', W, u_id, f_y, f_p, dfun(w_r, params_r)))
    g_u = -np.einsum('ijklm,m->i', W, residuals * np.einsum('ijklm,j,k,l->m', W, f_y, f_p, f_r))
    return np.concatenate(([g_y, g_p, g_r], g_u))
