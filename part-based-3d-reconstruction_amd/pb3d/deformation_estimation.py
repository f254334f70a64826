"""Part-wise deformation of voxel coordinates + re-projection (notebook-3 loop, BASELINE config 5).

The reference keeps this arithmetic in closures of its ipywidgets viewer
(reference utils/deformation_estimation.py: deform_coords :70-98, update :100-146, save_params :262-286,
save_deformed_grid :288-313); there is no importable function to replace, so this module exposes the
closures' numeric content as functions with the same argument meaning.  The widget plumbing itself is
UI and out of scope.  All array work runs on the device (csrc/deform.hip, points.hip, project.hip).
"""
import ctypes as C

import numpy as np

from . import _lib
from .camera_estimation import compute_partwise_iou
from .projection_utils import project_colored_voxels
from .voxel_utils import get_voxel_points_by_parts

__all__ = ["deform_coords", "deform_part", "evaluate_part_deform", "evaluate_part_deform_batch", "build_deformed_grid"]


def _scalars(image_shape, voxel_shape, deform):
    """the five float64 scalars of one_pass (:71-81), formed with Python floats exactly as upstream"""
    H_img, W_img = image_shape
    D, H, W = voxel_shape
    pix2vox_x = W / float(W_img)
    pix2vox_y = H / float(H_img)
    pix2vox_z = D / float(W_img)
    return (float(deform["scale_xz"]), float(deform["scale_y"]), deform["shift_xz"] * pix2vox_x, deform["shift_y"] * pix2vox_y,
            deform["shift_xz"] * pix2vox_z)


def deform_coords(coords, image_shape, voxel_shape, deform):
    """Unique deformed integer coordinates, lexicographically sorted (x,y,z) int64 rows -- what
    np.unique(np.vstack(seven jittered passes), axis=0) returns upstream.  `coords` must hold voxel
    indices (integer-valued), which is what get_voxel_points_by_parts produces."""
    p = np.ascontiguousarray(coords, np.float32)
    if p.ndim != 2 or p.shape[1] != 3:
        raise ValueError("coords must be (N,3)")
    if len(p) == 0:
        raise ValueError("cannot deform an empty point set")
    sxz, sy, kx, ky, kz = _scalars(image_shape, voxel_shape, deform)
    lib, ctx = _lib.load(), _lib.ctx()
    n = C.c_int64(0)
    _lib.check(lib.pb3d_deform_count(ctx, p.ctypes.data_as(C.POINTER(C.c_float)), len(p), sxz, sy, kx, ky, kz, C.byref(n)))
    out = np.empty((n.value, 3), np.int64)
    _lib.check(lib.pb3d_deform_fill(ctx, n.value, out.ctypes.data_as(_lib.i64p)))
    return out


def deform_part(voxel_grid, part_labels, part, deform, image_shape):
    """In-bounds deformed coordinates of one part and the colours upstream pairs with them (:102-118)."""
    voxel_shape = voxel_grid.shape[:3]
    coords, colors = get_voxel_points_by_parts(voxel_grid, part_labels, [part])
    if len(coords) == 0:
        return np.zeros((0, 3), np.int64), np.zeros((0, 3), np.uint8)
    cd = deform_coords(coords, image_shape, voxel_shape, deform)
    valid = ((cd[:, 0] >= 0) & (cd[:, 0] < voxel_shape[2]) & (cd[:, 1] >= 0) & (cd[:, 1] < voxel_shape[1]) &
             (cd[:, 2] >= 0) & (cd[:, 2] < voxel_shape[0]))
    cd = cd[valid]
    reps = max(1, int(len(cd) / len(colors)) + 1)
    return cd, np.repeat(colors, repeats=reps, axis=0)[:len(cd)]


def evaluate_part_deform(voxel_grid, part_labels, part, deform, image, cam_params):
    """Projection of the deformed part with the fixed camera and its IoU against `image` (:262-284)."""
    cd, cols = deform_part(voxel_grid, part_labels, part, deform, image.shape[:2])
    proj = project_colored_voxels(cd.astype(np.float32), cols, cam_params["cam_pos"], cam_params["target"],
                                  cam_params["f"], cam_params["cx"], cam_params["cy"], H=image.shape[0], W=image.shape[1])
    per, _ = compute_partwise_iou(proj, image, {part: part_labels[part]})
    return proj, float(per[part])


def evaluate_part_deform_batch(voxel_grid, part_labels, part, deforms, image, cam_params, stride=1):
    """[evaluate_part_deform(...)[1] for deform in deforms] -- the IoU of every deform tuple of a grid search (the loop the
    reference sketches at :148-258, `project_fast` with its point stride) -- with the part's points, the image and the camera
    resident and ONE pair of launches for the whole list.  Returns (ious, nvalid): ious[k] is the float the one-at-a-time
    path returns; nvalid[k] == 0 marks tuples whose deformed voxels all leave the grid (upstream prints a notice and has no
    IoU; the entry is 0.0)."""
    from . import device as dev
    from .camera_estimation import CameraObjective
    from .projection_utils import camera_args
    grid = _lib.as_u8(voxel_grid, "voxel_grid")
    img = _lib.as_u8(image, "image")
    H, W = img.shape[:2]
    A0, A1, A2 = grid.shape[:3]
    deforms = list(deforms)
    K = len(deforms)
    coords, colors = get_voxel_points_by_parts(grid, part_labels, [part])
    coords = np.ascontiguousarray(coords[::stride]); colors = colors[::stride]
    if K == 0:
        return [], np.zeros(0, np.int64)
    if len(coords) == 0:
        raise ValueError("cannot deform an empty point set")
    d5 = np.ascontiguousarray([_scalars((H, W), (A0, A1, A2), d) for d in deforms], np.float64).reshape(K, 5)
    # coords_def.astype(np.float32) is what upstream projects: float32 points, the camera's own dtypes
    _, _, R, cam, prec = camera_args(np.zeros((1, 3), np.float32), cam_params["cam_pos"], cam_params["target"], cam_params["f"],
                                     cam_params["cx"], cam_params["cy"])
    c = np.zeros(1, CameraObjective._CAM)
    c["R"][0] = R.reshape(9); c["cam"][0] = cam; c["f"][0] = float(cam_params["f"]); c["cx"][0] = float(cam_params["cx"])
    c["cy"][0] = float(cam_params["cy"]); c["prec"][0] = list(prec)
    color = np.ascontiguousarray(np.asarray(part_labels[part]).astype(np.uint8))
    inter = np.zeros(K, np.int64); uni = np.zeros(K, np.int64); nvalid = np.zeros(K, np.int64)
    d_pts = dev.from_numpy(coords); d_img = dev.from_numpy(img)
    try:
        _lib.check(_lib.load().pb3d_deform_iou_batch_dev(_lib.ctx(), C.c_void_p(d_pts.ptr), len(coords), _lib.p_dbl(d5), K, A0, A1, A2,
                                                         c.ctypes.data_as(C.c_void_p), H, W, C.c_void_p(d_img.ptr), _lib.p_u8(color),
                                                         inter.ctypes.data_as(_lib.i64p), uni.ctypes.data_as(_lib.i64p),
                                                         nvalid.ctypes.data_as(_lib.i64p)))
    finally:
        d_pts.free(); d_img.free()
    return [float(i / u) if (u > 0 and v > 0) else 0.0 for i, u, v in zip(inter, uni, nvalid)], nvalid


def build_deformed_grid(voxel_grid, part_labels, saved_params, image_shape):
    """Every saved part painted into a zero grid, in part_labels order (save_deformed_grid, :288-313).
    A part's points all carry the part colour, so the grid[z,y,x] = colour assignment is done by one
    device kernel per part straight from the jittered evaluation (no sort / dedup needed)."""
    from . import device as dev
    grid = _lib.as_u8(voxel_grid, "voxel_grid")
    if grid.ndim != 4 or grid.shape[3] != 3:
        raise ValueError("voxel_grid must be (A0,A1,A2,3)")
    A0, A1, A2 = grid.shape[:3]
    lib, ctx = _lib.load(), _lib.ctx()
    d_out = dev.DeviceBuffer(grid.size)
    d_out.zero()
    try:
        for part in part_labels:
            if part not in saved_params:
                continue
            coords, colors = get_voxel_points_by_parts(grid, part_labels, [part])
            if len(coords) == 0:
                continue
            sxz, sy, kx, ky, kz = _scalars(image_shape, (A0, A1, A2), saved_params[part]["deform"])
            d_pts = dev.from_numpy(coords)
            rgb = np.ascontiguousarray(colors[0], np.uint8)
            _lib.check(lib.pb3d_deform_paint_dev(ctx, C.c_void_p(d_pts.ptr), len(coords), sxz, sy, kx, ky, kz, A0, A1, A2,
                                                 _lib.p_u8(rgb), C.c_void_p(d_out.ptr)))
            dev.sync()
            d_pts.free()
        return d_out.download(grid.shape)
    finally:
        d_out.free()
