"""Z-buffer visibility of a voxel grid under a pinhole camera (row N5); host mirror of
reference utils/eval_helpers_intra.py:134-163 (compute_global_depth_buffer) and :168-190 (project_part_visible).
Upstream walks millions of points in Python `for` loops; here it is one atomicMin z-buffer kernel and one
compare kernel (csrc/project.hip)."""
import ctypes as C

import numpy as np

from . import _lib
from .projection_utils import _promotes_to_f64, camera_args

__all__ = ["compute_global_depth_buffer", "project_part_visible"]


def compute_global_depth_buffer(voxel_grid, cam, H, W):
    """(H,W) float32 depth of the nearest occupied voxel per pixel, +inf where none."""
    from . import device as dev
    grid = _lib.as_u8(voxel_grid, "voxel_grid")
    if grid.ndim != 4:
        raise ValueError("voxel_grid must be (A0,A1,A2,3)")
    A0, A1, A2, Cc = grid.shape
    lib, ctx = _lib.load(), _lib.ctx()
    d_grid = dev.from_numpy(grid)
    d_z = dev.DeviceBuffer(int(H) * int(W) * 4)
    bufs = [d_grid, d_z]
    try:
        n = C.c_int64(0)
        _lib.check(lib.pb3d_points_count_dev(ctx, C.c_void_p(d_grid.ptr), A0, A1, A2, Cc, None, 0, 1, C.byref(n)))
        d_pts = dev.DeviceBuffer(max(1, n.value) * 12); d_pc = dev.DeviceBuffer(max(1, n.value) * Cc)
        bufs += [d_pts, d_pc]
        _lib.check(lib.pb3d_points_fill_dev(ctx, C.c_void_p(d_grid.ptr), A0, A1, A2, Cc, None, 0, 1, n.value, C.c_void_p(d_pts.ptr),
                                            C.c_void_p(d_pc.ptr)))
        _, _, R, cp, prec = camera_args(np.zeros((1, 3), np.float32), cam["cam_pos"], cam["target"], cam["f"], cam["cx"], cam["cy"])
        _lib.check(lib.pb3d_depth_buffer_dev(ctx, C.c_void_p(d_pts.ptr), 0, n.value, _lib.p_dbl(R), _lib.p_dbl(cp), float(cam["f"]),
                                             float(cam["cx"]), float(cam["cy"]), prec, int(H), int(W), C.c_void_p(d_z.ptr)))
        return d_z.download((int(H), int(W)), np.float32)
    finally:
        for b in bufs:
            b.free()


def project_part_visible(pts3d, cam, zbuf, H, W, eps=1e-3):
    """(H,W) bool: pixels where some point of the part lies on the depth buffer (within eps)."""
    from . import device as dev
    p, pf64, R, cp, prec = camera_args(pts3d, cam["cam_pos"], cam["target"], cam["f"], cam["cx"], cam["cy"])
    zb = np.ascontiguousarray(zbuf, np.float32)
    if zb.shape != (int(H), int(W)):
        raise IndexError("zbuf shape does not match (H, W)")
    eps_f32 = int((not prec[0]) and not _promotes_to_f64(eps))
    d_p = dev.from_numpy(p) if len(p) else None
    d_zb = dev.from_numpy(zb); d_m = dev.DeviceBuffer(int(H) * int(W))
    try:
        _lib.check(_lib.load().pb3d_visible_mask_dev(_lib.ctx(), None if d_p is None else C.c_void_p(d_p.ptr), pf64, len(p), _lib.p_dbl(R),
                                                     _lib.p_dbl(cp), float(cam["f"]), float(cam["cx"]), float(cam["cy"]), prec,
                                                     C.c_void_p(d_zb.ptr), int(H), int(W), float(eps), eps_f32, C.c_void_p(d_m.ptr)))
        return d_m.download((int(H), int(W))).astype(bool)
    finally:
        for b in (d_p, d_zb, d_m):
            if b is not None:
                b.free()
