"""CPU ORACLE -- NumPy-signature front end of oracle/libpb3d_oracle.so.

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  Function names and
argument meaning follow the reference's L2 surface so parity tests read like calls into
the reference:

    reference utils/voxel_carving_utils.py : carve_voxel_grid_with_masks (:76),
        process_voxel_grid (:104), apply_colored_mask_to_voxel_grid (:128),
        part_carve (:139), global_carve (:269), _occupancy (:32), _mask_to_wh (:19)
    reference utils/voxel_utils.py         : get_voxel_points_by_parts (:7),
        voxel_grid_to_points (:35)
    reference utils/camera_geometry.py     : look_at_rotation (:3)
    reference utils/projection_utils.py    : project_colored_voxels (:5)
    reference utils/camera_estimation.py   : compute_partwise_iou (:770)

Parity pin: tests/test_oracle_golden.py checks every function below against vectors
captured from the imported reference (tools/gen_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PART_COLORS = {  # reference utils/config.py:29-40 (palette values are data the path keys on)
    "full_building": (253, 248, 96), "chhatris": (1, 220, 5), "plinth": (63, 138, 173),
    "dome": (190, 0, 255), "front_minarets": (0, 0, 255), "back_minarets": (5, 223, 223),
    "small_minarets": (255, 180, 80), "main_door": (180, 140, 255), "windows": (255, 120, 230),
    "background": (216, 224, 251),
}

_u8p = C.POINTER(C.c_uint8)
_i64 = C.c_int64


def build(force=False):
    if os.environ.get("PB3D_ORACLE_LIB"):       # a sanitizer build of the same file (make -C oracle SAN=1; README "Sanitizers")
        return os.environ["PB3D_ORACLE_LIB"]
    so = os.path.join(_HERE, "libpb3d_oracle.so")
    src = os.path.join(_HERE, "pb3d_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_points.restype = _i64
    return _LIB


def set_threads(n):
    lib().orc_set_threads(int(n))


def get_threads():
    return int(lib().orc_get_threads())


def _p(a, t=_u8p):
    return a.ctypes.data_as(t)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8c(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8:
        raise TypeError("oracle handles uint8 grids only, got %s" % a.dtype)
    return a


def _truth(a):
    return np.ascontiguousarray(np.asarray(a) != 0).view(np.uint8)


def mask_to_wh(mask, W, H):
    """_mask_to_wh, voxel_carving_utils.py:19-28: the (H,W) test is made first."""
    mask = np.asarray(mask)
    if mask.shape[:2] == (H, W):
        return mask.T
    if mask.shape[:2] == (W, H):
        return mask
    raise ValueError(f"Mask shape {mask.shape} incompatible with (W,H)=({W},{H})")


def rotation_matrix_inv(angle):
    M = np.empty(9, np.float64)
    if lib().orc_rotinv(int(angle), _dp(M)) != 0:
        raise ValueError("angle outside 0..90")
    return M.reshape(3, 3)


def affine_offset(M, shape):
    M = np.ascontiguousarray(M, np.float64)
    sh = (_i64 * 3)(*[int(s) for s in shape])
    off = np.empty(3, np.float64)
    lib().orc_offset(_dp(M), sh, _dp(off))
    return off


def affine_transform_u8(grid, M, off):
    g = _u8c(grid)
    W, H, D = g.shape
    out = np.empty_like(g)
    M = np.ascontiguousarray(M, np.float64)
    off = np.ascontiguousarray(off, np.float64)
    lib().orc_affine_u8(_p(g), _i64(W), _i64(H), _i64(D), _dp(M), _dp(off), _p(out))
    return out


def occupancy(grid):
    g = _u8c(grid)
    out = np.empty(g.shape[:3], np.uint8)
    lib().orc_occupancy(_p(g), _i64(out.size), _p(out))
    return out


def carve_voxel_grid_with_masks(voxel_grid, combined_mask):
    g = _u8c(voxel_grid)
    W, H, D = g.shape[:3]
    Cc = 3 if g.ndim == 4 else 1
    if g.ndim == 4 and g.shape[3] != 3:
        raise ValueError("colour grids must have 3 channels")
    m = mask_to_wh(combined_mask, W, H)
    if m.ndim == 2:
        mc = 1
    elif m.ndim == 3 and m.shape[2] == 3:
        # reference :90-95 broadcasts a (W,H,1,1) selector against a (W,H,D) channel slab,
        # which NumPy rejects for every non-degenerate shape -> the call ends in ValueError.
        raise ValueError("operands could not be broadcast together (RGB mask branch)")
    else:
        raise ValueError("Unsupported mask shape")
    mt = _truth(m)
    out = np.empty_like(g)
    rc = lib().orc_carve_mask(_p(g), _i64(W), _i64(H), _i64(D), Cc, _p(mt), mc, _p(out))
    assert rc == 0
    return out


def process_voxel_grid(voxel_grid, combined_mask, angle_interval=90):
    g = _u8c(voxel_grid)
    W, H, D = g.shape
    m = mask_to_wh(combined_mask, W, H)
    if m.ndim != 2:
        raise ValueError("oracle: process_voxel_grid takes a binary 2-D mask")
    mt = _truth(m)
    out = np.empty_like(g)
    rc = lib().orc_process_grid(_p(g), _i64(W), _i64(H), _i64(D), _p(mt), int(angle_interval), _p(out))
    if rc != 0:
        raise ValueError("angle_interval must be a positive integer")
    return out


def apply_colored_mask_to_voxel_grid(carved_voxel_grid, colored_mask):
    cv = _u8c(carved_voxel_grid)
    W, H, D = cv.shape
    rgb = _u8c(colored_mask)
    assert rgb.shape == (H, W, 3)
    out = np.empty((W, H, D, 3), np.uint8)
    lib().orc_color_apply(_p(cv), _i64(W), _i64(H), _i64(D), _p(rgb), _p(out))
    return out


def global_carve(binary_mask, semantic_mask_exterior, angle_interval=90, stride=4, visualize=False):
    b = _truth(binary_mask)
    h, w = b.shape
    rgb = _u8c(semantic_mask_exterior)
    assert rgb.shape == (h, w, 3)
    out = np.empty((w, h, w, 3), np.uint8)
    rc = lib().orc_global_carve(_p(b), _p(rgb), _i64(h), _i64(w), int(angle_interval), _p(out))
    if rc != 0:
        raise ValueError("angle_interval must be a positive integer")
    return out


def part_masks(semantic_mask, names, part_colors=PART_COLORS):
    """mask2d of voxel_carving_utils.py:143-146, (H,W) bool."""
    sm = np.asarray(semantic_mask)
    m = np.zeros(sm.shape[:2], bool)
    for n in names:
        c = np.asarray(part_colors[n])
        m |= (sm[..., 0] == c[0]) & (sm[..., 1] == c[1]) & (sm[..., 2] == c[2])
    return m


def part_carve(colored_grid, semantic_mask, group_jobs, visualize=False, part_colors=PART_COLORS):
    g = _u8c(colored_grid)
    W, H, D, _ = g.shape
    nj = len(group_jobs)
    msub = np.zeros((max(nj, 1), W, H), np.uint8)
    mcarve = np.zeros((max(nj, 1), W, H), np.uint8)
    ang = (C.c_int * max(nj, 1))()
    skip = (C.c_int * max(nj, 1))()
    for j, (names, angle) in enumerate(group_jobs):
        m2 = part_masks(semantic_mask, names, part_colors)
        skip[j] = 0 if m2.any() else 1
        ang[j] = int(angle)
        m = m2.T.astype(np.uint8)
        msub[j] = m
        mcarve[j] = np.ascontiguousarray(mask_to_wh(m, W, H))
    out = np.empty_like(g)
    rc = lib().orc_part_carve(_p(g), _i64(W), _i64(H), _i64(D), _p(msub), _p(mcarve), ang, skip, nj, _p(out))
    if rc != 0:
        raise ValueError("part_carve failed rc=%d" % rc)
    return out


def _points(grid, colors, stride):
    g = _u8c(grid)
    A0, A1, A2 = g.shape[:3]
    Cc = g.shape[3] if g.ndim == 4 else 1
    cols = np.ascontiguousarray(np.asarray(colors, np.uint8).reshape(-1, 3)) if colors is not None else np.zeros((0, 3), np.uint8)
    args = (_p(g), _i64(A0), _i64(A1), _i64(A2), Cc, _p(cols) if len(cols) else None, len(cols), int(stride))
    n = lib().orc_points(*args, None, None)
    pts = np.empty((n, 3), np.float32)
    pc = np.empty((n, Cc), np.uint8)
    lib().orc_points(*args, pts.ctypes.data_as(C.POINTER(C.c_float)), _p(pc))
    return pts, pc


def get_voxel_points_by_parts(grid, part_colors, part_names):
    cols = [part_colors[n] for n in part_names]
    if not cols:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)
    return _points(grid, cols, 1)


def voxel_grid_to_points(grid, axis="z", colormap="viridis", stride=2):
    grid = np.asarray(grid)
    W, H, D = grid.shape[:3]
    is_color = grid.ndim == 4 and grid.shape[3] == 3
    pts, pc = _points(grid, None, stride)
    if not is_color:
        raise NotImplementedError("oracle: colormap branch is visualisation-only (matplotlib)")
    return pts, pc, (H, W, D)


def look_at_rotation(eye, target, up=None):
    """camera_geometry.py:3-14, NumPy operations in the same order and dtypes."""
    if up is None:
        up = np.array([0, 1, 0], dtype=np.float32)
    z = target - eye
    z = z / np.linalg.norm(z)
    if np.allclose(np.abs(np.dot(z, up)), 1.0):
        up = np.array([0, 0, 1], dtype=np.float32)
    x = np.cross(up, z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=0)


def _is_f64_scalar(v):
    """NumPy-2 (NEP 50) promotion: Python floats/ints are weak; NumPy scalars/arrays carry a dtype."""
    if isinstance(v, (np.generic, np.ndarray)):
        return np.result_type(v, np.float32) == np.float64
    return False


def project_colored_voxels(pts3d, colors, cam_pos, target, f, cx, cy, H, W):
    pts3d = np.asarray(pts3d)
    cam_pos = np.asarray(cam_pos)
    target = np.asarray(target)
    R = look_at_rotation(cam_pos, target)
    t0 = int(np.result_type(pts3d, cam_pos, R) == np.float64)
    tm = int(t0 or _is_f64_scalar(f))
    tu = int(tm or _is_f64_scalar(cx))
    tv = int(tm or _is_f64_scalar(cy))
    prec = (C.c_int * 4)(t0, tm, tu, tv)
    if pts3d.dtype == np.float64:
        p = np.ascontiguousarray(pts3d, np.float64); pf64 = 1
    else:
        p = np.ascontiguousarray(pts3d, np.float32); pf64 = 0
    cols = np.ascontiguousarray(colors, np.uint8)
    n = p.shape[0]
    Rd = np.ascontiguousarray(R, np.float64)
    cd = np.ascontiguousarray(cam_pos, np.float64)
    img = np.empty((H, W, 3), np.uint8)
    lib().orc_project(p.ctypes.data_as(C.c_void_p), pf64, _p(cols), _i64(n), _dp(Rd), _dp(cd),
                      C.c_double(float(f)), C.c_double(float(cx)), C.c_double(float(cy)), prec,
                      int(H), int(W), _p(img))
    return img


def partwise_iou_counts(projected_img, image, colors):
    a = _u8c(projected_img); b = _u8c(image)
    cols = np.ascontiguousarray(np.asarray(colors, np.uint8).reshape(-1, 3))
    inter = np.zeros(len(cols), np.int64); uni = np.zeros(len(cols), np.int64)
    lib().orc_partwise_iou(_p(a), _p(b), _i64(a.size // 3), _p(cols), len(cols),
                           inter.ctypes.data_as(C.POINTER(_i64)), uni.ctypes.data_as(C.POINTER(_i64)))
    return inter, uni


def compute_partwise_iou(proj_mask, gt_mask, part_colors):
    """camera_estimation.py:770-787: ({part: inter/union or 0.0}, mean)."""
    names = list(part_colors.keys())
    inter, uni = partwise_iou_counts(proj_mask, gt_mask, [part_colors[n] for n in names])
    per = {}
    for n, i, u in zip(names, inter, uni):
        per[n] = (i / u) if u > 0 else 0.0
    return per, np.mean(list(per.values()))


# ---- A18: notebook-3 deformation closures (reference utils/deformation_estimation.py) -----------------
def deform_coords(coords, image_shape, voxel_shape, deform):
    """:70-98 -- unique, lexicographically sorted int64 rows of the 7-jitter deformation."""
    p = np.ascontiguousarray(coords, np.float32)
    H_img, W_img = image_shape
    D, H, W = voxel_shape
    kx = deform["shift_xz"] * (W / float(W_img))
    ky = deform["shift_y"] * (H / float(H_img))
    kz = deform["shift_xz"] * (D / float(W_img))
    out = np.empty((7 * len(p), 3), np.int64)
    lib().orc_deform_coords.restype = _i64
    m = lib().orc_deform_coords(p.ctypes.data_as(C.POINTER(C.c_float)), _i64(len(p)), C.c_double(deform["scale_xz"]),
                                C.c_double(deform["scale_y"]), C.c_double(kx), C.c_double(ky), C.c_double(kz),
                                out.ctypes.data_as(C.POINTER(_i64)))
    return out[:m].copy()


def deform_part(voxel_grid, part_labels, part, deform, image_shape):
    """:102-118 / :265-275: in-bounds deformed coordinates of one part and the colours paired with them."""
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
    """save_params, :262-284: projection of the deformed part and its IoU against the image."""
    cd, cols = deform_part(voxel_grid, part_labels, part, deform, image.shape[:2])
    proj = project_colored_voxels(cd.astype(np.float32), cols, cam_params["cam_pos"], cam_params["target"],
                                  cam_params["f"], cam_params["cx"], cam_params["cy"], image.shape[0], image.shape[1])
    per, _ = compute_partwise_iou(proj, image, {part: part_labels[part]})
    return proj, float(per[part])


def build_deformed_grid(voxel_grid, part_labels, saved_params, image_shape):
    """save_deformed_grid, :288-313: every saved part painted into a zero grid, in part_labels order."""
    out = np.zeros_like(voxel_grid, dtype=np.uint8)
    for part in part_labels:
        if part not in saved_params:
            continue
        cd, cols = deform_part(voxel_grid, part_labels, part, saved_params[part]["deform"], image_shape)
        if cd.size == 0:
            continue
        out[cd[:, 2], cd[:, 1], cd[:, 0]] = cols
    return out


# ---- N1/N2: component-guided carve, extrusion, recolouring, partwise_carve -----------------------------
def label6(mask):
    """scipy.ndimage.label(mask) (default 6-connected structure) -> (int32 labels, n)."""
    m = np.ascontiguousarray(np.asarray(mask) != 0).view(np.uint8)
    lab = np.zeros(m.shape, np.int32)
    lib().orc_label6.restype = _i64
    n = lib().orc_label6(_p(m), _i64(m.shape[0]), _i64(m.shape[1]), _i64(m.shape[2]), lab.ctypes.data_as(C.POINTER(C.c_int32)))
    return lab, int(n)


def left_right_guided_carve(colored_grid, semantic_mask, target_color, angle=60, visualize=False, stride=2):
    """voxel_carving_utils.py:163-210 (prints included: they are part of the notebook output)."""
    carved = colored_grid.copy()
    mask2d = np.all(semantic_mask == target_color, axis=-1)
    if not np.any(mask2d):
        print(f"[SKIP] No mask for color {target_color}")
        return carved
    lab, n = label6(np.all(colored_grid == target_color, axis=-1))
    print(f"[{target_color}] 3D components: {n}")
    for i in range(1, n + 1):
        m3 = lab == i
        idx = np.argwhere(m3)
        x0, y0, z0 = idx.min(axis=0)
        x1, y1, z1 = idx.max(axis=0) + 1
        print(f"  - Component {i}: bbox ({x0},{y0},{z0}) → ({x1},{y1},{z1})")
        crop2d = mask2d[y0:y1, x0:x1]
        sub = colored_grid[x0:x1, y0:y1, z0:z1].copy()
        occ = np.any(sub > 0, axis=-1).astype(np.uint8)
        carved_occ = process_voxel_grid(occ, crop2d, angle)
        print(f"    carved voxels: {np.count_nonzero(carved_occ)}")
        cc = sub * carved_occ[:, :, :, None]
        view = carved[x0:x1, y0:y1, z0:z1]
        view[m3[x0:x1, y0:y1, z0:z1]] = 0
        keep = np.any(cc > 0, axis=-1)
        view[keep] = cc[keep]
    return carved


def extrude_from_surface(grid, mask_2d, axis, direction="+", depth=5, fill_color=None):
    """voxel_carving_utils.py:213-248."""
    occ = occupancy(grid)
    W, H, D = occ.shape
    filled = np.zeros(occ.shape, bool)
    if axis == 2:
        start = np.argmax(occ if direction == "+" else occ[:, :, ::-1], axis=2)
        if direction == "-":
            start = D - 1 - start
        valid = np.asarray(mask_2d).T
        for d in range(depth):
            z = start + d if direction == "+" else start - d
            ok = (z >= 0) & (z < D) & valid
            xs, ys = np.nonzero(ok)
            filled[xs, ys, z[xs, ys]] = True
    elif axis == 0:
        start = np.argmax(occ if direction == "+" else occ[::-1], axis=0)
        if direction == "-":
            start = W - 1 - start
        valid = np.asarray(mask_2d)
        for d in range(depth):
            x = start + d if direction == "+" else start - d
            ok = (x >= 0) & (x < W) & valid
            ys, zs = np.nonzero(ok)
            filled[x[ys, zs], ys, zs] = True
    out = grid.copy()
    out[filled] = 0 if fill_color is None else fill_color
    return out


def recolor_backward_components(voxel_grid, color, new_color, k=4, sort_axis=2):
    """voxel_carving_utils.py:252-266."""
    lab, n = label6(np.all(voxel_grid == color, axis=-1))
    comps = [(i, np.argwhere(lab == i)[:, sort_axis].mean()) for i in range(1, n + 1)]
    keep = {i for i, _ in sorted(comps, key=lambda t: t[1])[:k]}
    out = voxel_grid.copy()
    for i in range(1, n + 1):
        if i not in keep:
            out[lab == i] = new_color
    return out


def partwise_carve(colored_voxel_grid, semantic_mask_exterior, semantic_mask_full, part_colors_np, group_jobs, part_symmetry,
                   extrusion_depths, recolor_back_minarets=True, visualize=False, stride=4):
    """voxel_carving_utils.py:302-400."""
    grid = part_carve(colored_voxel_grid, semantic_mask_exterior, group_jobs)
    for part, angle in part_symmetry.items():
        grid = left_right_guided_carve(grid, semantic_mask_exterior, part_colors_np[part], angle=angle)
    for part, depth in extrusion_depths.items():
        mk = np.all(semantic_mask_full == part_colors_np[part], axis=-1)
        for ax, dr in ((2, "+"), (2, "-"), (0, "+"), (0, "-")):
            grid = extrude_from_surface(grid, mk, axis=ax, direction=dr, depth=depth, fill_color=part_colors_np[part])
    if recolor_back_minarets:
        oriented = np.flip(grid.transpose(2, 1, 0, 3), axis=1)
        grid = recolor_backward_components(oriented, part_colors_np["front_minarets"], new_color=part_colors_np["back_minarets"],
                                           k=2, sort_axis=0)
    return grid


# ---- N4: camera objective; N5: z-buffer visibility --------------------------------------------------------
def mask_parts_from_image(image, part_colors, selected_parts):
    """mask_utils.py:89-97: the image with everything but the selected parts' colours blacked out."""
    img = np.asarray(image)
    out = np.zeros_like(img)
    for part in selected_parts:
        c = np.asarray(part_colors[part])
        hit = (img[..., 0] == c[0]) & (img[..., 1] == c[1]) & (img[..., 2] == c[2])
        out[hit] = c
    return out


def camera_objective(voxel_pts, voxel_colors, seg_img, selected_labels, p, H, W):
    """evaluate() of launch_smart_aligner, camera_estimation.py:597-603: minus the mean part IoU."""
    proj = project_colored_voxels(voxel_pts, voxel_colors, p["cam_pos"], p["target"], p["f"], p["cx"], p["cy"], H, W)
    _, iou = compute_partwise_iou(proj, seg_img, selected_labels)
    return -iou


def _pin_args(pts3d, cam):
    pts3d = np.asarray(pts3d)
    cam_pos = np.asarray(cam["cam_pos"]); target = np.asarray(cam["target"])
    R = look_at_rotation(cam_pos, target)
    t0 = int(np.result_type(pts3d, cam_pos, R) == np.float64)
    tm = int(t0 or _is_f64_scalar(cam["f"]))
    tu = int(tm or _is_f64_scalar(cam["cx"])); tv = int(tm or _is_f64_scalar(cam["cy"]))
    prec = (C.c_int * 4)(t0, tm, tu, tv)
    pf64 = int(pts3d.dtype == np.float64)
    p = np.ascontiguousarray(pts3d, np.float64 if pf64 else np.float32)
    return p, pf64, np.ascontiguousarray(R, np.float64), np.ascontiguousarray(cam_pos, np.float64), prec, t0


def compute_global_depth_buffer(voxel_grid, cam, H, W):
    """eval_helpers_intra.py:134-163."""
    pts, _, _ = voxel_grid_to_points(voxel_grid, stride=1)
    p, pf64, R, cp, prec, _ = _pin_args(pts, cam)
    zbuf = np.empty((H, W), np.float32)
    lib().orc_depth_buffer(p.ctypes.data_as(C.c_void_p), pf64, _i64(len(p)), _dp(R), _dp(cp), C.c_double(float(cam["f"])),
                           C.c_double(float(cam["cx"])), C.c_double(float(cam["cy"])), prec, int(H), int(W),
                           zbuf.ctypes.data_as(C.POINTER(C.c_float)))
    return zbuf


def project_part_visible(pts3d, cam, zbuf, H, W, eps=1e-3):
    """eval_helpers_intra.py:168-190."""
    p, pf64, R, cp, prec, t0 = _pin_args(pts3d, cam)
    zb = np.ascontiguousarray(zbuf, np.float32)
    mask = np.zeros((H, W), np.uint8)
    eps_f32 = int((not t0) and not _is_f64_scalar(eps))
    lib().orc_visible_mask(p.ctypes.data_as(C.c_void_p), pf64, _i64(len(p)), _dp(R), _dp(cp), C.c_double(float(cam["f"])),
                           C.c_double(float(cam["cx"])), C.c_double(float(cam["cy"])), prec, zb.ctypes.data_as(C.POINTER(C.c_float)),
                           int(H), int(W), C.c_double(float(eps)), eps_f32, _p(mask))
    return mask.astype(bool)


# =====================================================================================================
# Grids of any dtype SciPy's interpolation takes (bool, int8..int64, uint8..uint64, float32/64, complex64/128) through the rotate + carve
# loop -- reference utils/voxel_carving_utils.py:104-126 does not look at the dtype: scipy.ndimage.affine_transform(order=1,
# mode="constant", cval=0) returns the input's dtype, carve_voxel_grid_with_masks (np.where) keeps it.  A NumPy restatement (float64 array
# operations are IEEE and unfused, in the order of the scalar form above), pinned against SciPy itself in tests/test_oracle_golden.py.
#   value:  acc = (((v00*wx0)*wz0 + (v01*wx0)*wz1) + (v10*wx1)*wz0) + (v11*wx1)*wz1, taps with an exactly-zero weight skipped
#   store (ni_interpolation.c, CASE_INTERP_OUT*): floats (T)acc; unsigned: acc > 0 ? acc + 0.5 : 0, clipped to [0, MAX], truncated;
#           signed: acc > 0 ? acc + 0.5 : acc - 0.5, clipped to [MIN, MAX], truncated; bool: (unsigned char)acc (truncation) != 0 ...
#           complex: real and imaginary parts separately (scipy/ndimage/_interpolation.py splits them)
# =====================================================================================================
TYPED_DTYPES = ("bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "complex64", "complex128")


def _store_typed(acc, dtype):
    dt = np.dtype(dtype)
    if dt.kind == "f":
        return acc.astype(dt)
    if dt.kind == "b":
        return acc.astype(np.uint8).astype(bool)          # C cast double -> unsigned char: truncation (acc is within [0, 1])
    info = np.iinfo(dt)
    if dt.kind == "u":
        t = np.where(acc > 0, acc + 0.5, 0.0)
        t = np.where(t > float(info.max), float(info.max), t)
        t = np.where(t < 0, 0.0, t)
    else:
        t = np.where(acc > 0, acc + 0.5, acc - 0.5)
        t = np.where(t > float(info.max), float(info.max), t)
        t = np.where(t < float(info.min), float(info.min), t)
    with np.errstate(invalid="ignore"):
        if dt.itemsize == 8:
            # the C cast of a double that equals 2^63 (int64) / 2^64 (uint64) after the clip is out of range; on x86-64 it yields INT64_MIN /
            # (for the unsigned conversion sequence) 2^63 ... SciPy's result, whatever the platform gives -- test data stay below 2^62
            return np.trunc(t).astype(dt)
        return np.trunc(t).astype(np.int64).astype(dt)


def affine_transform_typed(grid, M, off):
    g = np.ascontiguousarray(grid)
    if g.dtype.name not in TYPED_DTYPES:
        raise RuntimeError("data type not supported")
    if g.dtype.kind == "c":
        out = np.empty_like(g)
        out.real = affine_transform_typed(np.ascontiguousarray(g.real), M, off)
        out.imag = affine_transform_typed(np.ascontiguousarray(g.imag), M, off)
        return out
    W, H, D = g.shape
    M = np.asarray(M, np.float64).reshape(3, 3); off = np.asarray(off, np.float64)
    x = np.arange(W, dtype=np.float64)[:, None]; z = np.arange(D, dtype=np.float64)[None, :]
    def coord(ma, mb, mc, o):
        c = 0.0 + x * ma
        c = c + 0.0 * mb
        c = c + z * mc
        return c + o
    cc0 = coord(M[0, 0], M[0, 1], M[0, 2], off[0]); cc2 = coord(M[2, 0], M[2, 1], M[2, 2], off[2])
    inside = ~((cc0 < 0.0) | (cc0 > W - 1) | (cc2 < 0.0) | (cc2 > D - 1))
    f0 = np.floor(np.where(inside, cc0, 0.0)); f2 = np.floor(np.where(inside, cc2, 0.0))
    s0 = f0.astype(np.int64); s2 = f2.astype(np.int64)
    wx0 = 1.0 - (np.where(inside, cc0, 0.0) - f0); wx1 = 1.0 - wx0
    wz0 = 1.0 - (np.where(inside, cc2, 0.0) - f2); wz1 = 1.0 - wz0
    s0b = np.minimum(s0 + 1, W - 1); s2b = np.minimum(s2 + 1, D - 1)      # (a tap beyond the last index only ever has weight 0)
    gd = g.astype(np.float64)
    acc = np.zeros((W, H, D), np.float64)
    def tap(a, b, wa, wb):
        v = gd[a[:, None, :], np.arange(H)[None, :, None], b[:, None, :]]
        prod = (v * wa[:, None, :]) * wb[:, None, :]
        live = ((wa != 0.0) & (wb != 0.0))[:, None, :]
        return np.where(live, prod, 0.0), live
    for (a, b, wa, wb) in ((s0, s2, wx0, wz0), (s0, s2b, wx0, wz1), (s0b, s2, wx1, wz0), (s0b, s2b, wx1, wz1)):
        pr, live = tap(a, b, wa, wb)
        acc = np.where(live, acc + pr, acc)
    acc = np.where(inside[:, None, :], acc, 0.0)
    return _store_typed(acc, g.dtype)


def process_voxel_grid_typed(voxel_grid, combined_mask, angle_interval=90):
    """process_voxel_grid (reference :104-126) for any dtype SciPy's interpolation takes."""
    g = np.ascontiguousarray(voxel_grid)
    W, H, D = g.shape
    m = mask_to_wh(combined_mask, W, H)
    keep = (np.asarray(m) != 0)[:, :, None]
    for angle in range(0, 91, int(angle_interval)):
        Minv = rotation_matrix_inv(angle)
        g = affine_transform_typed(g, Minv, affine_offset(Minv, (W, H, D)))
        g = np.where(keep, g, 0)          # (as upstream: a bool grid comes back as int64 -- NumPy's promotion of the Python 0)
    return g
