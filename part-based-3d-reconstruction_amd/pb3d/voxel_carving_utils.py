"""Orthographic semantic voxel carving on MI355X -- host mirror of the reference interface.

Same function names, argument meaning, return shapes/dtypes and error behaviour as
reference utils/voxel_carving_utils.py; every array operation runs in hand-written HIP
kernels (csrc/*.hip) through the ctypes C-ABI (include/pb3d.h).  Grids are uint8 with axes
(W=x, H=y, D=z[,3]).  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from .config import PART_COLORS, PART_COLORS_NP  # noqa: F401  (re-exported like upstream)

__all__ = ["carve_voxel_grid_with_masks", "process_voxel_grid", "apply_colored_mask_to_voxel_grid", "part_carve",
           "global_carve"]

# optional visualisation hook (reference utils/visualization.plot_voxel); set by pb3d.install()
plot_voxel = None


def _mask_to_wh(mask, W, H):
    """Orient a 2-D/3-D mask as (W,H[,3]); reference :19-28.  The (H,W) test comes first, so a
    square mask is always transposed."""
    mask = np.asarray(mask)
    if mask.shape[:2] == (H, W):
        return mask.T
    if mask.shape[:2] == (W, H):
        return mask
    raise ValueError(f"Mask shape {mask.shape} incompatible with (W,H)=({W},{H})")


def _rotation_matrix_inv(angle):
    """inv of the Y-rotation by `angle` degrees, bit-identical to the reference's LAPACK result
    for the integer angles 0..90 its loops can produce (reference :65-69; pinned table)."""
    M = np.empty((3, 3), np.float64)
    if int(angle) != angle:
        raise ValueError("only integer angles 0..90 are pinned")
    _lib.check(_lib.load().pb3d_rotinv(int(angle), _lib.p_dbl(M)))
    return M


def _occupancy(grid):
    """any(grid > 0, axis=-1) as uint8; reference :32-33."""
    g = _lib.as_u8(grid, "grid")
    if g.ndim != 4 or g.shape[3] != 3:
        raise ValueError("_occupancy expects a (W,H,D,3) grid")
    out = np.empty(g.shape[:3], np.uint8)
    _lib.check(_lib.load().pb3d_occupancy(_lib.ctx(), _lib.p_u8(g), out.size, _lib.p_u8(out)))
    return out


def carve_voxel_grid_with_masks(voxel_grid, combined_mask):
    """np.where(mask[x,y], grid, 0) broadcast over z (and channel); reference :76-97."""
    g = _lib.as_u8(voxel_grid, "voxel_grid")
    if g.ndim not in (3, 4) or (g.ndim == 4 and g.shape[3] != 3):
        raise ValueError(f"voxel_grid must be (W,H,D) or (W,H,D,3), got {g.shape}")
    W, H, D = g.shape[:3]
    mask = _mask_to_wh(combined_mask, W, H)
    if mask.ndim == 2:
        m = _lib.truth_u8(mask)
        out = np.empty_like(g)
        _lib.check(_lib.load().pb3d_carve_mask(_lib.ctx(), _lib.p_u8(g), W, H, D, 3 if g.ndim == 4 else 1,
                                               _lib.p_u8(m), _lib.p_u8(out)))
        return out
    if mask.ndim == 3 and mask.shape[2] == 3:
        # Upstream's RGB-mask branch (:90-95) selects with a (W,H,1,1) array against a (W,H,D)
        # channel slab; NumPy cannot broadcast that for any non-degenerate shape, so the call
        # always ends in this ValueError.  Mirrored rather than "fixed".
        raise ValueError(f"operands could not be broadcast together with shapes ({W},{H},1,1) ({W},{H},{D}) ()")
    raise ValueError("Unsupported mask shape")


def process_voxel_grid(voxel_grid, combined_mask, angle_interval=90):
    """for angle in range(0, 91, angle_interval): rotate about Y (trilinear, SciPy semantics) and
    carve; cumulative, never rotated back.  reference :104-126.  The loop runs on the device."""
    g = _lib.as_u8(voxel_grid, "voxel_grid")
    if g.ndim != 3:
        raise ValueError(f"process_voxel_grid rotates occupancy grids (W,H,D); got shape {g.shape}")
    W, H, D = g.shape
    if isinstance(angle_interval, (bool, np.bool_)) or not isinstance(angle_interval, (int, np.integer)):
        raise TypeError(f"'{type(angle_interval).__name__}' object cannot be interpreted as an integer")
    if angle_interval == 0:
        raise ValueError("range() arg 3 must not be zero")
    if angle_interval < 0:
        return g.copy()  # range(0, 91, negative) is empty: upstream returns the input grid
    mask = _mask_to_wh(combined_mask, W, H)
    if mask.ndim != 2:
        carve_voxel_grid_with_masks(g, combined_mask)  # raises what upstream raises
    m = _lib.truth_u8(mask)
    out = np.empty_like(g)
    _lib.check(_lib.load().pb3d_process_grid(_lib.ctx(), _lib.p_u8(g), W, H, D, _lib.p_u8(m), int(min(angle_interval, 91)),
                                             _lib.p_u8(out)))
    return out


def apply_colored_mask_to_voxel_grid(carved_voxel_grid, colored_mask):
    """out[x,y,z,:] = colored_mask[y,x,:] where carved == 1 else 0; reference :128-136."""
    cv = _lib.as_u8(carved_voxel_grid, "carved_voxel_grid")
    W, H, D = cv.shape
    rgb = np.ascontiguousarray(np.asarray(colored_mask).astype(np.uint8, copy=False))
    if rgb.shape != (H, W, 3):
        raise ValueError(f"colored_mask shape {rgb.shape} does not match (H,W,3)=({H},{W},3)")
    out = np.empty((W, H, D, 3), np.uint8)
    _lib.check(_lib.load().pb3d_color_apply(_lib.ctx(), _lib.p_u8(cv), W, H, D, _lib.p_u8(rgb), _lib.p_u8(out)))
    return out


def _job_masks(semantic_mask, group_jobs, W, H, part_colors):
    """Per job: (mask2d.T as uint8, what _mask_to_wh makes of it, angle, skip) -- reference :143-151."""
    sm = np.asarray(semantic_mask)
    nj = len(group_jobs)
    msub = np.zeros((max(nj, 1), W, H), np.uint8)
    mcarve = np.zeros((max(nj, 1), W, H), np.uint8)
    angles = (C.c_int * max(nj, 1))()
    skip = (C.c_int * max(nj, 1))()
    for j, (names, angle) in enumerate(group_jobs):
        sel = np.zeros(sm.shape[:2], bool)
        for n in names:
            sel |= np.all(sm == part_colors[n], axis=-1)
        skip[j] = 0 if sel.any() else 1
        angles[j] = int(angle)
        m = sel.T.astype(np.uint8)
        if m.shape != (W, H):
            raise ValueError(f"operands could not be broadcast together: mask {m.shape} vs grid ({W},{H})")
        msub[j] = m
        mcarve[j] = _mask_to_wh(m, W, H)
    return msub, mcarve, angles, skip


def part_carve(colored_grid, semantic_mask, group_jobs, visualize=False):
    """Per part group: select its pixels, carve the group's occupancy with its own symmetry
    angle, overlay the survivors; reference :139-160."""
    g = _lib.as_u8(colored_grid, "colored_grid")
    if g.ndim != 4 or g.shape[3] != 3:
        raise ValueError("colored_grid must be (W,H,D,3)")
    W, H, D, _ = g.shape
    msub, mcarve, angles, skip = _job_masks(semantic_mask, group_jobs, W, H, PART_COLORS)
    for j in range(len(group_jobs)):
        if not skip[j] and angles[j] <= 0:
            raise ValueError("range() arg 3 must not be zero" if angles[j] == 0 else "negative angle steps are not supported")
        angles[j] = min(angles[j], 91) if angles[j] > 0 else angles[j]
    out = np.empty_like(g)
    _lib.check(_lib.load().pb3d_part_carve(_lib.ctx(), _lib.p_u8(g), W, H, D, _lib.p_u8(msub), _lib.p_u8(mcarve), angles, skip,
                                           len(group_jobs), _lib.p_u8(out)))
    return out


def global_carve(binary_mask, semantic_mask_exterior, angle_interval=90, stride=4, visualize=False):
    """ones((w,h,w)) -> process_voxel_grid -> colours; reference :269-298.  Returns (w,h,w,3) uint8."""
    b = np.asarray(binary_mask)
    if b.ndim != 2:
        raise ValueError("not enough values to unpack" if b.ndim < 2 else "too many values to unpack (expected 2)")
    h, w = b.shape
    if isinstance(angle_interval, (bool, np.bool_)) or not isinstance(angle_interval, (int, np.integer)):
        raise TypeError(f"'{type(angle_interval).__name__}' object cannot be interpreted as an integer")
    if angle_interval == 0:
        raise ValueError("range() arg 3 must not be zero")
    rgb = np.ascontiguousarray(np.asarray(semantic_mask_exterior).astype(np.uint8, copy=False))
    if rgb.shape != (h, w, 3):
        raise ValueError(f"semantic mask shape {rgb.shape} does not match (h,w,3)=({h},{w},3)")
    if angle_interval < 0:
        # empty angle loop: the all-ones grid is coloured as is
        return apply_colored_mask_to_voxel_grid(np.ones((w, h, w), np.uint8), rgb)
    bt = _lib.truth_u8(b)
    out = np.empty((w, h, w, 3), np.uint8)
    _lib.check(_lib.load().pb3d_global_carve(_lib.ctx(), _lib.p_u8(bt), _lib.p_u8(rgb), h, w, int(min(angle_interval, 91)),
                                             _lib.p_u8(out)))
    if visualize and plot_voxel is not None:
        from .voxel_utils import voxel_grid_to_points
        pts, cols, _ = voxel_grid_to_points(out, stride=stride)
        if pts.shape[0] > 0:
            plot_voxel(pts, cols, title="After global symmetric carving")
    return out
