"""torch-facing operators over the C ABI (include/nlml_hpe.h).

Each function takes CUDA(HIP) tensors, checks shapes/dtypes on the host (a wrong shape must
never reach a hand-written kernel), and launches on torch's CURRENT stream.  The same entry
points are registered as ``torch.ops.nlml_hpe.*`` custom ops.  No CPU path:
a CPU tensor raises.
"""
from __future__ import annotations

import torch

from . import _lib

F_REF = 1404
LATENT = 9


def _stream_ptr(device=None) -> int:
    """torch's current HIP stream OF `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def _need_cuda(t: torch.Tensor, name: str, dtype) -> None:
    if not t.is_cuda:
        raise _lib.NlmlError(f"{name}: expected a GPU tensor (there is no CPU fallback), got device {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


class _on_device_of:
    """Every operand on ONE GPU, and the launch on THAT GPU: a HIP launch goes to the calling thread's current device, so a
    model living on cuda:1 called while cuda:0 is current would run on GPU 0 against GPU 1's memory, on a foreign stream.
    `with _on_device_of(x, blob, ...) as stream:` checks that all tensors share a device, makes it current for the launch and
    yields that device's current stream handle; outputs must be allocated on `x.device` by the caller."""

    def __init__(self, *named):
        ts = [(n, t) for n, t in named if t is not None]
        self.device = ts[0][1].device
        for n, t in ts[1:]:
            if t.device != self.device:
                raise _lib.NlmlError(f"{n} is on {t.device} but {ts[0][0]} is on {self.device}: all operands must share one GPU")
        self.guard = torch.cuda.device(self.device)

    def __enter__(self) -> int:
        self.guard.__enter__()
        return _stream_ptr(self.device)

    def __exit__(self, *exc):
        return self.guard.__exit__(*exc)


def normalize_ipd(raw: torch.Tensor, normalize: bool = True, return_valid: bool = False):
    """raw f32[B,468,3] -> features f32[B,1404] (FeatureExtractor.py:30-66,101), bit-exact."""
    _need_cuda(raw, "raw", torch.float32)
    if raw.dim() != 3 or raw.shape[1:] != (468, 3):
        raise ValueError(f"raw: expected [B,468,3], got {tuple(raw.shape)}")
    raw = raw.contiguous()
    B = raw.shape[0]
    out = torch.empty((B, F_REF), dtype=torch.float32, device=raw.device)
    valid = torch.empty((B,), dtype=torch.uint8, device=raw.device) if return_valid else None
    with _on_device_of(("raw", raw)) as stream:
        _lib.check(_lib.lib().nlml_normalize_ipd(raw.data_ptr(), B, int(bool(normalize)), out.data_ptr(),
                                                 valid.data_ptr() if valid is not None else None, stream),
                   "nlml_normalize_ipd")
    return (out, valid.bool()) if return_valid else out


def encoder_heads_fwd(x: torch.Tensor, blob: torch.Tensor, F: int, return_latent: bool = False,
                      return_valid: bool = False):
    """x f32[B,F] -> radians f32[B,3] (CombinedAnglePredictionModel.forward, Model_Builder.py:115-126)."""
    _need_cuda(x, "x", torch.float32)
    _need_cuda(blob, "blob", torch.uint8)
    if x.dim() != 2 or x.shape[1] != F:
        raise ValueError(f"x: expected [B,{F}], got {tuple(x.shape)}")
    if x.stride(1) != 1:
        x = x.contiguous()
    B = x.shape[0]
    ldx = x.stride(0) if B > 1 else F
    out = torch.empty((B, 3), dtype=torch.float32, device=x.device)
    latent = torch.empty((B, LATENT), dtype=torch.float32, device=x.device) if return_latent else None
    valid = torch.empty((B,), dtype=torch.uint8, device=x.device) if return_valid else None
    with _on_device_of(("x", x), ("blob", blob)) as stream:
        _lib.check(_lib.lib().nlml_encoder_heads_fwd(
            x.data_ptr(), ldx, B, F, blob.data_ptr(), blob.numel(), out.data_ptr(),
            latent.data_ptr() if latent is not None else None,
            valid.data_ptr() if valid is not None else None, stream), "nlml_encoder_heads_fwd")
    res = [out]
    if return_latent:
        res.append(latent)
    if return_valid:
        res.append(valid.bool())
    return res[0] if len(res) == 1 else tuple(res)


def encoder_heads_fwd_debug(x: torch.Tensor, blob: torch.Tensor, F: int, want_stamps: bool = False):
    """Diagnostic build: (out [B,3], latent [B,9], pre_tanh [B,64][, stamps i64[ceil(B/64),4,16]]); include/nlml_hpe.h."""
    _need_cuda(x, "x", torch.float32)
    _need_cuda(blob, "blob", torch.uint8)
    if x.dim() != 2 or x.shape[1] != F:
        raise ValueError(f"x: expected [B,{F}], got {tuple(x.shape)}")
    x = x.contiguous()
    B = x.shape[0]
    out = torch.empty((B, 3), dtype=torch.float32, device=x.device)
    latent = torch.empty((B, LATENT), dtype=torch.float32, device=x.device)
    pre = torch.empty((B, 64), dtype=torch.float32, device=x.device)
    stamps = torch.zeros(((B + 63) // 64, 4, 16), dtype=torch.int64, device=x.device) if want_stamps else None
    with _on_device_of(("x", x), ("blob", blob)) as stream:
        _lib.check(_lib.lib().nlml_encoder_heads_fwd_debug(x.data_ptr(), F, B, F, blob.data_ptr(), blob.numel(),
                                                           out.data_ptr(), latent.data_ptr(), pre.data_ptr(),
                                                           stamps.data_ptr() if want_stamps else None, stream),
                   "nlml_encoder_heads_fwd_debug")
    return (out, latent, pre, stamps) if want_stamps else (out, latent, pre)


def landmarks_to_pose(raw: torch.Tensor, blob: torch.Tensor, normalize: bool = True, return_latent: bool = False,
                      return_valid: bool = False):
    """Fused K1+K2: raw f32[B,468,3] -> radians f32[B,3]; normalised features never reach HBM."""
    _need_cuda(raw, "raw", torch.float32)
    _need_cuda(blob, "blob", torch.uint8)
    if raw.dim() != 3 or raw.shape[1:] != (468, 3):
        raise ValueError(f"raw: expected [B,468,3], got {tuple(raw.shape)}")
    raw = raw.contiguous()
    B = raw.shape[0]
    out = torch.empty((B, 3), dtype=torch.float32, device=raw.device)
    latent = torch.empty((B, LATENT), dtype=torch.float32, device=raw.device) if return_latent else None
    valid = torch.empty((B,), dtype=torch.uint8, device=raw.device) if return_valid else None
    with _on_device_of(("raw", raw), ("blob", blob)) as stream:
        _lib.check(_lib.lib().nlml_landmarks_to_pose(
            raw.data_ptr(), B, int(bool(normalize)), blob.data_ptr(), blob.numel(), out.data_ptr(),
            latent.data_ptr() if latent is not None else None,
            valid.data_ptr() if valid is not None else None, stream), "nlml_landmarks_to_pose")
    res = [out]
    if return_latent:
        res.append(latent)
    if return_valid:
        res.append(valid.bool())
    return res[0] if len(res) == 1 else tuple(res)


def tucker_objective(Wm: torch.Tensor, x: torch.Tensor, params: torch.Tensor, cos_params: torch.Tensor,
                     x_index: torch.Tensor | None = None, return_xhat: bool = False, order="reference"):
    """Batched objective (TD_Tester.py:31-58): Wm f32[135,1404], x f32[M,1404], params f64[N,8],
    cos_params f64[3,3,4] -> err f64[N] (+ x_hat f64[N,1404]).  order: "reference" (the default and the parity mode: np.einsum's
    operation order and numpy's pairwise sum -- the reference's bits) or "fast" (opt-in: a GEMM on the f64 matrix cores, <= 1e-12
    relative, ~5x the evaluations/s)."""
    order = _lib.td_order_from_name(order)
    _need_cuda(Wm, "Wm", torch.float32)
    _need_cuda(x, "x", torch.float32)
    _need_cuda(params, "params", torch.float64)
    _need_cuda(cos_params, "cos_params", torch.float64)
    if tuple(Wm.shape) != (135, F_REF):
        raise ValueError(f"Wm: expected [135,1404], got {tuple(Wm.shape)}")
    if x.dim() != 2 or x.shape[1] != F_REF:
        raise ValueError(f"x: expected [M,1404], got {tuple(x.shape)}")
    if params.dim() != 2 or params.shape[1] != 8:
        raise ValueError(f"params: expected [N,8], got {tuple(params.shape)}")
    if tuple(cos_params.shape) != (3, 3, 4):
        raise ValueError(f"cos_params: expected [3,3,4], got {tuple(cos_params.shape)}")
    Wm, x, params, cos_params = Wm.contiguous(), x.contiguous(), params.contiguous(), cos_params.contiguous()
    N = params.shape[0]
    if x_index is not None:
        _need_cuda(x_index, "x_index", torch.int32)
        if x_index.shape != (N,):
            raise ValueError("x_index: expected [N]")
        x_index = x_index.contiguous()
        if N and (int(x_index.min()) < 0 or int(x_index.max()) >= x.shape[0]):
            raise IndexError("x_index out of range")
    elif x.shape[0] != N:
        raise ValueError(f"x has {x.shape[0]} rows but params has {N} (pass x_index to share rows)")
    err = torch.empty((N,), dtype=torch.float64, device=x.device)
    xh = torch.empty((N, F_REF), dtype=torch.float64, device=x.device) if return_xhat else None
    with _on_device_of(("x", x), ("Wm", Wm), ("params", params), ("cos_params", cos_params), ("x_index", x_index)) as stream:
        _lib.check(_lib.lib().nlml_tucker_objective_ex(
            Wm.data_ptr(), x.data_ptr(), F_REF, x_index.data_ptr() if x_index is not None else None,
            params.data_ptr(), cos_params.data_ptr(), N, err.data_ptr(),
            xh.data_ptr() if xh is not None else None, order, stream), "nlml_tucker_objective_ex")
    return (err, xh) if return_xhat else err


# ---------------------------------------------------------------------------------------------
# torch.ops.nlml_hpe.* (SURVEY.md 8b "Underlying op") come from COMPILED code: csrc/torch_ops.cpp, a TORCH_LIBRARY shim over the same
# C ABI (shape / dtype checks, torch's allocator, the operand device's current stream), built next to this file as
# libnlml_torch_ops.so by csrc/Makefile.  Registered: normalize_ipd, encoder_heads_fwd, landmarks_to_pose, encoder_heads_fwd_small,
# landmarks_to_pose_small (explicit workspace), landmarks_to_pose_valid (pose + face mask: the video tick's forward), tucker_objective, tucker_powell, video_post, cosine_table.  GPU backend only: a CPU
# tensor has no kernel to dispatch to and raises.  A missing library raises here -- the ops are part of the boundary.
import os as _os

TORCH_OPS_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libnlml_torch_ops.so")


def _load_torch_ops():
    if not _os.path.exists(TORCH_OPS_PATH):
        raise _lib.NlmlError(f"{TORCH_OPS_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(make -C nlml_hpe_amd/csrc); torch.ops.nlml_hpe.* are registered from it")
    _lib.lib()                                  # the C ABI library the shim links against, checked symbol by symbol first
    torch.ops.load_library(TORCH_OPS_PATH)
    for name in ("normalize_ipd", "encoder_heads_fwd", "landmarks_to_pose", "encoder_heads_fwd_small", "landmarks_to_pose_small",
                 "landmarks_to_pose_valid", "tucker_objective", "tucker_powell", "video_post", "cosine_table"):
        getattr(torch.ops.nlml_hpe, name)       # AttributeError if the library did not register it


_load_torch_ops()


def tucker_powell(Wm: torch.Tensor, x: torch.Tensor, cos_params: torch.Tensor, x0: torch.Tensor | None = None, order="reference"):
    """Batched Test() (TD_Tester.py:162-199): one Powell minimisation per row of x, on device.  order as in tucker_objective:
    "reference" (default, the parity mode) walks scipy's own trajectory on the reference's objective bits; "fast" (opt-in, ~3x the
    faces/s) minimises the matrix-core objective: same algorithm, but the flat minimum makes the END POINT sensitive to the last bits
    of the objective -- 6e-3 deg from scipy on clean grid faces (FX5), and on BASELINE config 3's noisy faces median 1.8e-3 deg (per face, largest of the three angles),
    10 % of the faces > 0.02 deg, 0.3 % > 1 deg, max 8.7 deg (bench.py extra.td_powell_fast_order reports it live).

    Returns dict(x=f64[N,8] (w_y,w_p,w_r radians + u_id), fun=f64[N], nfev=i32[N], nit=i32[N], status=i32[N]).
    """
    _need_cuda(Wm, "Wm", torch.float32)
    _need_cuda(x, "x", torch.float32)
    _need_cuda(cos_params, "cos_params", torch.float64)
    if tuple(Wm.shape) != (135, F_REF):
        raise ValueError(f"Wm: expected [135,1404], got {tuple(Wm.shape)}")
    if x.dim() != 2 or x.shape[1] != F_REF:
        raise ValueError(f"x: expected [N,1404], got {tuple(x.shape)}")
    if tuple(cos_params.shape) != (3, 3, 4):
        raise ValueError(f"cos_params: expected [3,3,4], got {tuple(cos_params.shape)}")
    Wm, x, cos_params = Wm.contiguous(), x.contiguous(), cos_params.contiguous()
    N = x.shape[0]
    if x0 is not None:
        _need_cuda(x0, "x0", torch.float64)
        if tuple(x0.shape) != (N, 8):
            raise ValueError(f"x0: expected [{N},8], got {tuple(x0.shape)}")
        x0 = x0.contiguous()
    order = _lib.td_order_from_name(order)
    dev = x.device
    res = torch.empty((N, 8), dtype=torch.float64, device=dev)
    fun = torch.empty((N,), dtype=torch.float64, device=dev)
    nfev = torch.empty((N,), dtype=torch.int32, device=dev)
    nit = torch.empty((N,), dtype=torch.int32, device=dev)
    status = torch.empty((N,), dtype=torch.int32, device=dev)
    with _on_device_of(("x", x), ("Wm", Wm), ("cos_params", cos_params), ("x0", x0)) as stream:
        _lib.check(_lib.lib().nlml_tucker_powell_ex(Wm.data_ptr(), x.data_ptr(), F_REF, cos_params.data_ptr(), N,
                                                    x0.data_ptr() if x0 is not None else None, res.data_ptr(), fun.data_ptr(),
                                                    nfev.data_ptr(), nit.data_ptr(), status.data_ptr(), order, stream),
                   "nlml_tucker_powell_ex")
    return {"x": res, "fun": fun, "nfev": nfev, "nit": nit, "status": status}


def cosine_table(angles_rad: torch.Tensor, cos_params: torch.Tensor) -> torch.Tensor:
    """U[i,j] = a_j*cos(b_j*w_i + c_j) + d_j in f64 (NLML_HPE_MLPHeadsTrainer.py:71-73,179-205).

    angles_rad f32[n] (radians), cos_params f64[R,4] rows (a,b,c,d) -> f64[n,R]."""
    _need_cuda(angles_rad, "angles_rad", torch.float32)
    _need_cuda(cos_params, "cos_params", torch.float64)
    if angles_rad.dim() != 1 or cos_params.dim() != 2 or cos_params.shape[1] != 4:
        raise ValueError(f"expected angles [n] and cos_params [R,4], got {tuple(angles_rad.shape)}, {tuple(cos_params.shape)}")
    angles_rad, cos_params = angles_rad.contiguous(), cos_params.contiguous()
    n, R = angles_rad.shape[0], cos_params.shape[0]
    out = torch.empty((n, R), dtype=torch.float64, device=angles_rad.device)
    with _on_device_of(("angles_rad", angles_rad), ("cos_params", cos_params)) as stream:
        _lib.check(_lib.lib().nlml_cosine_table(angles_rad.data_ptr(), n, cos_params.data_ptr(), R, out.data_ptr(), stream),
                   "nlml_cosine_table")
    return out


def mode5_product(core: torch.Tensor, U_feat: torch.Tensor) -> torch.Tensor:
    """W = core x_5 U_feat (TD_main.py:232-238): core f32[..., R5], U_feat f32[M, R5] -> W f32[..., M]."""
    _need_cuda(core, "core", torch.float32)
    _need_cuda(U_feat, "U_feat", torch.float32)
    if U_feat.dim() != 2 or core.dim() < 1 or core.shape[-1] != U_feat.shape[1]:
        raise ValueError(f"core [..., R5] and U_feat [M, R5] disagree: {tuple(core.shape)}, {tuple(U_feat.shape)}")
    lead = tuple(core.shape[:-1])
    c2 = core.contiguous().reshape(-1, core.shape[-1])
    U_feat = U_feat.contiguous()
    Q, R5, M = c2.shape[0], c2.shape[1], U_feat.shape[0]
    W = torch.empty((Q, M), dtype=torch.float32, device=core.device)
    with _on_device_of(("core", core), ("U_feat", U_feat)) as stream:
        _lib.check(_lib.lib().nlml_mode5_product(c2.data_ptr(), U_feat.data_ptr(), Q, R5, M, W.data_ptr(), stream),
                   "nlml_mode5_product")
    return W.reshape(lead + (M,))


_small_ws: dict = {}


def _small_workspace(B: int, F: int, device) -> torch.Tensor:
    """Scratch for the layer-per-launch path (its contents never matter).  Buffers are cached per device and stream and NEVER
    released or replaced: a captured hipGraph keeps replaying into the pointer it was captured with, so growing means
    adding a larger buffer next to the old one.  The first one is sized for 4,096 faces of the reference width (46 MB)."""
    need = _lib.lib().nlml_encoder_heads_small_workspace_bytes(B, F)
    # one pool per (device, stream): launches on different streams may overlap and must not share scratch
    pool = _small_ws.setdefault((str(device), _stream_ptr(device)), [])
    for ws in pool:
        if ws.numel() >= need:
            return ws
    size = max(need, _lib.lib().nlml_encoder_heads_small_workspace_bytes(4096, F_REF))
    ws = torch.empty((size,), dtype=torch.uint8, device=device)
    pool.append(ws)
    return ws


def encoder_heads_fwd_small(x: torch.Tensor, blob: torch.Tensor, F: int, return_latent: bool = False,
                            return_valid: bool = False, workspace: torch.Tensor | None = None):
    """encoder_heads_fwd for small batches (split-f16 blob only): one launch per big layer (five launches; seven with a strict blob: its
    tail is two launches and the f32 re-evaluation launch follows) spread over the whole chip, bit-identical results."""
    _need_cuda(x, "x", torch.float32)
    _need_cuda(blob, "blob", torch.uint8)
    if x.dim() != 2 or x.shape[1] != F:
        raise ValueError(f"x: expected [B,{F}], got {tuple(x.shape)}")
    if x.stride(1) != 1:
        x = x.contiguous()
    B = x.shape[0]
    ldx = x.stride(0) if B > 1 else F
    ws = workspace if workspace is not None else _small_workspace(B, F, x.device)
    out = torch.empty((B, 3), dtype=torch.float32, device=x.device)
    latent = torch.empty((B, LATENT), dtype=torch.float32, device=x.device) if return_latent else None
    valid = torch.empty((B,), dtype=torch.uint8, device=x.device) if return_valid else None
    with _on_device_of(("x", x), ("blob", blob), ("workspace", ws)) as stream:
        _lib.check(_lib.lib().nlml_encoder_heads_fwd_small(
            x.data_ptr(), ldx, B, F, blob.data_ptr(), blob.numel(), out.data_ptr(),
            latent.data_ptr() if latent is not None else None,
            valid.data_ptr() if valid is not None else None, ws.data_ptr(), ws.numel(), stream),
            "nlml_encoder_heads_fwd_small")
    res = [out]
    if return_latent:
        res.append(latent)
    if return_valid:
        res.append(valid.bool())
    return res[0] if len(res) == 1 else tuple(res)


def landmarks_to_pose_small(raw: torch.Tensor, blob: torch.Tensor, normalize: bool = True, return_latent: bool = False,
                            return_valid: bool = False, workspace: torch.Tensor | None = None):
    """landmarks_to_pose for small batches (split-f16 blob only): one launch per big layer (five launches; seven with a strict blob) spread
    over the whole chip, bit-identical results."""
    _need_cuda(raw, "raw", torch.float32)
    _need_cuda(blob, "blob", torch.uint8)
    if raw.dim() != 3 or raw.shape[1:] != (468, 3):
        raise ValueError(f"raw: expected [B,468,3], got {tuple(raw.shape)}")
    raw = raw.contiguous()
    B = raw.shape[0]
    ws = workspace if workspace is not None else _small_workspace(B, F_REF, raw.device)
    out = torch.empty((B, 3), dtype=torch.float32, device=raw.device)
    latent = torch.empty((B, LATENT), dtype=torch.float32, device=raw.device) if return_latent else None
    valid = torch.empty((B,), dtype=torch.uint8, device=raw.device) if return_valid else None
    with _on_device_of(("raw", raw), ("blob", blob), ("workspace", ws)) as stream:
        _lib.check(_lib.lib().nlml_landmarks_to_pose_small(
            raw.data_ptr(), B, int(bool(normalize)), blob.data_ptr(), blob.numel(), out.data_ptr(),
            latent.data_ptr() if latent is not None else None,
            valid.data_ptr() if valid is not None else None, ws.data_ptr(), ws.numel(), stream),
            "nlml_landmarks_to_pose_small")
    res = [out]
    if return_latent:
        res.append(latent)
    if return_valid:
        res.append(valid.bool())
    return res[0] if len(res) == 1 else tuple(res)


_streamed_ws: dict = {}


def landmarks_to_pose_streamed(raw: torch.Tensor, blob: torch.Tensor, normalize: bool = True, return_latent: bool = False,
                               return_valid: bool = False, workspace: torch.Tensor | None = None):
    """landmarks_to_pose (strict-fast blob only) as trunk launch + streamed tail launch + the f32 re-evaluation launch
    (nlml_landmarks_to_pose_streamed): bit-identical to the fused kernel, measured 1.4-2.2 % faster at 65,536 faces; nothing calls it by default
    (DESIGN.md section 3).  The hand-over buffer (1 KB per face) is cached per device and stream like the layer-per-launch path's scratch."""
    _need_cuda(raw, "raw", torch.float32)
    _need_cuda(blob, "blob", torch.uint8)
    if raw.dim() != 3 or raw.shape[1:] != (468, 3):
        raise ValueError(f"raw: expected [B,468,3], got {tuple(raw.shape)}")
    raw = raw.contiguous()
    B = raw.shape[0]
    if workspace is None:
        need = max(16, _lib.lib().nlml_encoder_heads_workspace_bytes(B, F_REF))
        pool = _streamed_ws.setdefault((str(raw.device), _stream_ptr(raw.device)), [])
        workspace = next((w for w in pool if w.numel() >= need), None)
        if workspace is None:
            workspace = torch.empty((need,), dtype=torch.uint8, device=raw.device)
            pool.append(workspace)
    out = torch.empty((B, 3), dtype=torch.float32, device=raw.device)
    latent = torch.empty((B, LATENT), dtype=torch.float32, device=raw.device) if return_latent else None
    valid = torch.empty((B,), dtype=torch.uint8, device=raw.device) if return_valid else None
    with _on_device_of(("raw", raw), ("blob", blob), ("workspace", workspace)) as stream:
        _lib.check(_lib.lib().nlml_landmarks_to_pose_streamed(
            raw.data_ptr(), B, int(bool(normalize)), blob.data_ptr(), blob.numel(), out.data_ptr(),
            latent.data_ptr() if latent is not None else None,
            valid.data_ptr() if valid is not None else None, workspace.data_ptr(), workspace.numel(), stream),
            "nlml_landmarks_to_pose_streamed")
    res = [out]
    if return_latent:
        res.append(latent)
    if return_valid:
        res.append(valid.bool())
    return res[0] if len(res) == 1 else tuple(res)
