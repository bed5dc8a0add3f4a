"""Oracle: IPD landmark normalisation (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``Read_Landmarks_and_Normalizing_using_IPD``
(/root/reference/helpers/FeatureExtractor.py:30-66) and the f32 cast its three
callers apply (FeatureExtractor.py:101,142,187: ``torch.tensor(list).float()``).

Arithmetic contract (what the HIP kernel must reproduce bit for bit):
  * landmark coordinates are f32 values read as Python floats => exact f64;
  * ipd = ||lm[33] - lm[263]||_2 in f64 via ``np.linalg.norm`` (:45), replaced
    by 1e-6 when exactly 0 (:47-48);
  * if normalize: (v - lm[1][c]) / ipd, both in f64 (:55-61), c = coordinate;
  * flatten x,y,z interleaved (:63), round once to f32.
"""
from __future__ import annotations

import numpy as np

NOSE_IDX = 1        # FeatureExtractor.py:85  ref_point = landmark[1]
LEFT_EYE_IDX = 33   # FeatureExtractor.py:35
RIGHT_EYE_IDX = 263  # FeatureExtractor.py:36


def ipd_f64(lm: np.ndarray) -> np.ndarray:
    """f64[B] inter-pupil distance of f32/f64 landmarks [B,468,3] (:38-48)."""
    lm = np.asarray(lm)
    out = np.empty(lm.shape[0], dtype=np.float64)
    for b in range(lm.shape[0]):
        left = lm[b, LEFT_EYE_IDX].astype(np.float64)
        right = lm[b, RIGHT_EYE_IDX].astype(np.float64)
        v = np.linalg.norm(left - right)          # same call as the reference (:45)
        out[b] = 1e-6 if v == 0 else v            # :47-48
    return out


def normalize_ipd(lm: np.ndarray, normalize: bool = True) -> np.ndarray:
    """raw landmarks f32[B,468,3] -> features f32[B,1404] (vectorised restatement)."""
    lm = np.asarray(lm, dtype=np.float32)
    if lm.ndim == 2:
        lm = lm[None]
    b = lm.shape[0]
    v = lm.astype(np.float64)
    if normalize:
        ref = v[:, NOSE_IDX:NOSE_IDX + 1, :]       # ref_list (:85-86)
        ipd = ipd_f64(lm)[:, None, None]
        v = (v - ref) / ipd                         # :55-61, f64 subtract then f64 divide
    return v.reshape(b, -1).astype(np.float32)      # .float() at :101


def normalize_ipd_loop(lm_face: np.ndarray, normalize: bool = True) -> np.ndarray:
    """One face, scalar Python loop in the reference's statement order (small cases only)."""
    lm_face = np.asarray(lm_face, dtype=np.float32)
    left = np.array([float(c) for c in lm_face[LEFT_EYE_IDX]])
    right = np.array([float(c) for c in lm_face[RIGHT_EYE_IDX]])
    ipd = np.linalg.norm(left - right)
    if ipd == 0:
        ipd = 1e-6
    ref = [float(c) for c in lm_face[NOSE_IDX]]
    out = []
    for p in lm_face:
        x, y, z = float(p[0]), float(p[1]), float(p[2])
        if normalize:
            x -= ref[0]; y -= ref[1]; z -= ref[2]
            x /= ipd; y /= ipd; z /= ipd
        out.extend([x, y, z])
    return np.array(out, dtype=np.float64).astype(np.float32)


def no_face_mask(features: np.ndarray) -> np.ndarray:
    """bool[B]: all-zero feature row == 'no face detected' sentinel.

    FeatureExtractor.py:105-106 returns zeros(1404) when FaceMesh finds nothing;
    callers skip such rows (NLML_HPE_Test.py:257-260, generatePose_on_video.py:193-196).
    """
    return (np.asarray(features) == 0).all(axis=1)
