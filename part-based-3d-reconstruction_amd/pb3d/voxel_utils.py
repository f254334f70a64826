"""Grid -> points (ordered stream compaction on the device); host mirror of
reference utils/voxel_utils.py:7-21 and :35-51."""
import ctypes as C

import numpy as np

from . import _hostmem, _lib

__all__ = ["get_voxel_points_by_parts", "voxel_grid_to_points"]


def _compact(grid, colors, stride):
    g = _lib.as_u8(grid, "grid")
    A0, A1, A2 = g.shape[:3]
    Cc = g.shape[3] if g.ndim == 4 else 1
    if colors is not None and len(colors):
        cols = np.ascontiguousarray(np.asarray(colors, np.int64).reshape(-1, 3))
        # a palette entry outside 0..255 can never equal a uint8 voxel
        ok = np.all((cols >= 0) & (cols <= 255), axis=1)
        cols = np.ascontiguousarray(cols[ok].astype(np.uint8))
        if len(cols) == 0:
            return np.zeros((0, 3), np.float32), np.zeros((0, Cc), np.uint8)
        if len(cols) > 32:
            raise ValueError("at most 32 part colours per call")
        cptr, nc = _lib.p_u8(cols), len(cols)
    else:
        cptr, nc = None, 0
    lib, ctx = _lib.load(), _lib.ctx()
    n = C.c_int64(0)
    _lib.check(lib.pb3d_points_count(ctx, _lib.p_u8(g), A0, A1, A2, Cc, cptr, nc, int(stride), C.byref(n)))
    pts = _hostmem.empty((n.value, 3), np.float32)
    pc = _hostmem.empty((n.value, Cc), np.uint8)
    _lib.check(lib.pb3d_points_fill(ctx, n.value, pts.ctypes.data_as(C.POINTER(C.c_float)), _lib.p_u8(pc)))
    return pts, pc


def get_voxel_points_by_parts(grid, part_colors, part_names):
    """Points (x,y,z) = (a2,a1,a0) float32 and colours of every voxel whose RGB equals one of the
    named parts' colours, in numpy.where order; reference :7-21."""
    grid = np.asarray(grid)
    if grid.ndim != 4 or grid.shape[3] != 3:
        raise ValueError("grid must be (A0,A1,A2,3)")
    cols = [part_colors[name] for name in part_names]
    if not cols:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)
    return _compact(grid, cols, 1)


def voxel_grid_to_points(grid, axis="z", colormap="viridis", stride=2):
    """Occupied voxels on the [::stride] lattice as points*stride (+ colours); reference :35-51.
    Returns (pts, colors, (H, W, D))."""
    grid = np.asarray(grid)
    W, H, D = grid.shape[:3]
    is_color = grid.ndim == 4 and grid.shape[3] == 3
    if grid.ndim == 4 and not is_color:
        raise ValueError("too many values to unpack (expected 3)")
    pts, pc = _compact(grid, None, stride)
    if is_color:
        return pts, pc, (H, W, D)
    # occupancy grids are coloured by a matplotlib colormap along one axis (visualisation only)
    import matplotlib.pyplot as plt
    xs, ys, zs = (pts[:, k] / np.float32(stride) for k in range(3))
    with np.errstate(divide="ignore", invalid="ignore"):
        vals = {"x": xs, "y": ys, "z": zs}[axis].astype(np.int64) / {"x": W - 1, "y": H - 1, "z": D - 1}[axis]
    colors = (plt.get_cmap(colormap)(vals)[:, :3] * 255).astype(np.uint8)
    return pts, colors, (H, W, D)
