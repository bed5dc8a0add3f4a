"""ctypes loader for the plain-C oracle (oracle/csrc/oracle.c).  TEST INFRASTRUCTURE."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def normalize_ipd(raw: np.ndarray, normalize: bool = True) -> np.ndarray:
    raw = np.ascontiguousarray(raw, dtype=np.float32).reshape(-1, 1404)
    out = np.empty_like(raw)
    lib().oracle_normalize_ipd(_p(raw), C.c_int64(raw.shape[0]), C.c_int(int(normalize)), _p(out))
    return out


def encoder_heads(x: np.ndarray, params, order: int = 2, want_latent=False, want_pre_tanh=False):
    """params: oracle.encoder_heads.Params.  order 0 = k ascending, 1 = one MFMA-order chain per output, 2 (default) = the f32
    HIP kernel's order: MFMA-order chains with layers 0 to 3 summed in blocks of 128 k (encoder_heads.hip fold_block)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, F = x.shape
    ew = [np.ascontiguousarray(w) for w, _ in params.enc]
    eb = [np.ascontiguousarray(b) for _, b in params.enc]
    hw = [np.ascontiguousarray(w) for n in ("yaw", "pitch", "roll") for w, _ in params.heads[n]]
    hb = [np.ascontiguousarray(b) for n in ("yaw", "pitch", "roll") for _, b in params.heads[n]]
    arr = lambda xs: (C.c_void_p * len(xs))(*[a.ctypes.data for a in xs])
    out = np.empty((B, 3), np.float32)
    lat = np.empty((B, 9), np.float32) if want_latent else None
    pre = np.empty((B, 64), np.float32) if want_pre_tanh else None
    lib().oracle_encoder_heads_f32(_p(x), C.c_int64(B), C.c_int(F), arr(ew), arr(eb), arr(hw), arr(hb), C.c_int(order),
                                   _p(out), _p(lat) if lat is not None else None, _p(pre) if pre is not None else None)
    res = [out]
    if want_latent:
        res.append(lat)
    if want_pre_tanh:
        res.append(pre)
    return res[0] if len(res) == 1 else tuple(res)


def tucker_objective(Wm, x, params, cosp, want_xhat=False, device_order=False, reference_order=False):
    """reference_order: np.einsum's own (i,j,k,l)-outer sum of separately rounded products and numpy's pairwise np.sum --
    bit-identical to the reference's objective (FX4); device_order: the fast kernel's reduction tree."""
    Wm = np.ascontiguousarray(Wm, dtype=np.float32).reshape(135, 1404)
    x = np.ascontiguousarray(x, dtype=np.float32)
    params = np.ascontiguousarray(params, dtype=np.float64)
    cosp = np.ascontiguousarray(cosp, dtype=np.float64)
    N = params.shape[0]
    err = np.empty(N)
    xh = np.empty((N, 1404)) if want_xhat else None
    if reference_order:
        lib().oracle_tucker_objective_reforder(_p(Wm), _p(x), _p(params), _p(cosp), C.c_int64(N), _p(err),
                                               _p(xh) if want_xhat else None)
        return (err, xh) if want_xhat else err
    lib().oracle_tucker_objective(_p(Wm), _p(x), _p(params), _p(cosp), C.c_int64(N), _p(err), _p(xh) if want_xhat else None,
                                  C.c_int(int(device_order)))
    return (err, xh) if want_xhat else err
