"""Per-part IoU between a projected image and a mask; host mirror of
reference utils/camera_estimation.py:770-787 (the inner metric of every re-projection loop)."""
import ctypes as C

import numpy as np

from . import _lib

__all__ = ["compute_partwise_iou"]


def partwise_iou_counts(proj_mask, gt_mask, colors):
    a = _lib.as_u8(proj_mask, "proj_mask").reshape(-1, 3)
    b = _lib.as_u8(gt_mask, "gt_mask").reshape(-1, 3)
    if a.shape != b.shape:
        raise ValueError(f"operands could not be broadcast together with shapes {a.shape} {b.shape}")
    cols = np.asarray(colors, np.int64).reshape(-1, 3)
    inter = np.zeros(len(cols), np.int64)
    uni = np.zeros(len(cols), np.int64)
    ok = np.all((cols >= 0) & (cols <= 255), axis=1)
    c8 = np.ascontiguousarray(cols[ok].astype(np.uint8))
    if len(c8):
        i8 = np.zeros(len(c8), np.int64); u8 = np.zeros(len(c8), np.int64)
        for s in range(0, len(c8), 32):
            chunk = np.ascontiguousarray(c8[s:s + 32])
            _lib.check(_lib.load().pb3d_partwise_iou(_lib.ctx(), _lib.p_u8(a), _lib.p_u8(b), a.shape[0], _lib.p_u8(chunk),
                                                     len(chunk), i8[s:].ctypes.data_as(_lib.i64p), u8[s:].ctypes.data_as(_lib.i64p)))
        inter[ok] = i8; uni[ok] = u8
    return inter, uni


def compute_partwise_iou(proj_mask, gt_mask, part_colors):
    """({part: inter/union, or 0.0 when the union is empty}, mean over parts)."""
    names = list(part_colors.keys())
    inter, uni = partwise_iou_counts(proj_mask, gt_mask, [part_colors[n] for n in names])
    per_part = {}
    for name, i, u in zip(names, inter, uni):
        per_part[name] = (i / u) if u > 0 else 0.0
    return per_part, np.mean(list(per_part.values()))
