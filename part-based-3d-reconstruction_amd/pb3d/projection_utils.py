"""Pinhole scatter projection of coloured voxels; host mirror of reference utils/projection_utils.py:5-23."""
import ctypes as C

import numpy as np

from . import _lib
from .camera_geometry import look_at_rotation, project  # noqa: F401

__all__ = ["project_colored_voxels"]


def _promotes_to_f64(v):
    """NumPy-2 promotion: Python scalars are weak, NumPy scalars / arrays carry their dtype."""
    if isinstance(v, (np.generic, np.ndarray)):
        return np.result_type(v, np.float32) == np.float64
    return False


def camera_args(pts3d, cam_pos, target, f, cx, cy):
    """Everything the projection kernels need from the caller's camera: the look-at rotation (host NumPy, same
    dtypes as upstream), the points in their own float width and the NumPy-2 promotion flags of each stage."""
    pts3d = np.asarray(pts3d)
    cam_pos = np.asarray(cam_pos)
    target = np.asarray(target)
    R = look_at_rotation(cam_pos, target)
    t0 = int(np.result_type(pts3d, cam_pos, R) == np.float64)
    tm = int(t0 or _promotes_to_f64(f))
    prec = (C.c_int * 4)(t0, tm, int(tm or _promotes_to_f64(cx)), int(tm or _promotes_to_f64(cy)))
    pf64 = int(pts3d.dtype == np.float64)
    return (np.ascontiguousarray(pts3d, np.float64 if pf64 else np.float32), pf64, np.ascontiguousarray(R, np.float64),
            np.ascontiguousarray(cam_pos, np.float64), prec)


def project_colored_voxels(pts3d, colors, cam_pos, target, f, cx, cy, H, W):
    """(H,W,3) uint8 image of the points seen from cam_pos looking at target; among points that
    land on one pixel the last in input order wins (NumPy fancy-assignment semantics)."""
    pts3d = np.asarray(pts3d)
    cam_pos = np.asarray(cam_pos)
    target = np.asarray(target)
    R = look_at_rotation(cam_pos, target)
    t0 = int(np.result_type(pts3d, cam_pos, R) == np.float64)
    tm = int(t0 or _promotes_to_f64(f))
    tu = int(tm or _promotes_to_f64(cx))
    tv = int(tm or _promotes_to_f64(cy))
    prec = (C.c_int * 4)(t0, tm, tu, tv)
    if pts3d.ndim != 2 or pts3d.shape[1] != 3:
        raise ValueError("pts3d must be (N,3)")
    pf64 = int(pts3d.dtype == np.float64)
    p = np.ascontiguousarray(pts3d, np.float64 if pf64 else np.float32)
    cols = np.ascontiguousarray(np.asarray(colors).astype(np.uint8, copy=False))
    if cols.shape != (p.shape[0], 3):
        raise ValueError("colors must be (N,3)")
    Rd = np.ascontiguousarray(R, np.float64)
    cd = np.ascontiguousarray(cam_pos, np.float64)
    img = np.empty((int(H), int(W), 3), np.uint8)
    _lib.check(_lib.load().pb3d_project(_lib.ctx(), p.ctypes.data_as(C.c_void_p), pf64, _lib.p_u8(cols), p.shape[0],
                                        _lib.p_dbl(Rd), _lib.p_dbl(cd), float(f), float(cx), float(cy), prec,
                                        int(H), int(W), _lib.p_u8(img)))
    return img
