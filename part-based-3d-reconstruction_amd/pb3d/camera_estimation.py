"""Per-part IoU between a projected image and a mask; host mirror of
reference utils/camera_estimation.py:770-787 (the inner metric of every re-projection loop)."""
import ctypes as C

import numpy as np

from . import _lib

__all__ = ["compute_partwise_iou", "CameraObjective", "projection_iou_by_part", "random_search", "coordinate_descent", "powell_search"]


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


class CameraObjective:
    """The objective of the camera aligner, reference utils/camera_estimation.py:597-603
    (`evaluate(p)` = minus the mean IoU of the selected parts between the projection and the part image),
    with the point cloud, its colours and the image RESIDENT in HBM: the random / coordinate / Powell loops
    (:606-725) re-project the same points hundreds of times and only the nine camera numbers change.

        obj = CameraObjective(voxel_pts, voxel_colors, seg_img, selected_labels)
        value = obj(p)                       # p: dict with cam_pos, target, f, cx, cy (H, W default to the image's)
        values = obj.evaluate_batch([p1, p2, ...])
    """

    def __init__(self, voxel_pts, voxel_colors, seg_img, selected_labels):
        from . import device as dev
        self._dev = dev
        pts = np.asarray(voxel_pts)
        self._pf64 = int(pts.dtype == np.float64)
        self._pts_dtype = pts.dtype
        self._pts_host = np.ascontiguousarray(pts, np.float64 if self._pf64 else np.float32)
        if self._pts_host.ndim != 2 or self._pts_host.shape[1] != 3:
            raise ValueError("voxel_pts must be (N,3)")
        cols = np.ascontiguousarray(np.asarray(voxel_colors).astype(np.uint8, copy=False))
        if cols.shape != (len(self._pts_host), 3):
            raise ValueError("voxel_colors must be (N,3)")
        self.n = len(self._pts_host)
        seg = _lib.as_u8(seg_img, "seg_img")
        self.H, self.W = seg.shape[:2]
        self.names = list(selected_labels.keys())
        self._colors = np.ascontiguousarray(np.array([selected_labels[k] for k in self.names], np.uint8).reshape(-1, 3))
        if len(self._colors) > 32:
            raise ValueError("at most 32 parts")
        self._d_pts = dev.from_numpy(self._pts_host) if self.n else None
        self._d_cols = dev.from_numpy(cols) if self.n else None
        self._d_seg = dev.from_numpy(seg)
        self._d_img = dev.DeviceBuffer(self.H * self.W * 3)

    def __call__(self, p):
        from .projection_utils import camera_args
        H, W = int(p.get("H", self.H)), int(p.get("W", self.W))
        if (H, W) != (self.H, self.W):
            raise ValueError("operands could not be broadcast together: projection and part image differ in size")
        # only dtypes matter for the promotion flags: a one-row stand-in avoids touching the resident points
        _, _, R, cam, prec = camera_args(np.zeros((1, 3), self._pts_dtype), p["cam_pos"], p["target"], p["f"], p["cx"], p["cy"])
        lib, ctx = _lib.load(), _lib.ctx()
        _lib.check(lib.pb3d_project_dev(ctx, None if not self.n else C.c_void_p(self._d_pts.ptr), self._pf64,
                                        None if not self.n else C.c_void_p(self._d_cols.ptr), self.n, _lib.p_dbl(R), _lib.p_dbl(cam),
                                        float(p["f"]), float(p["cx"]), float(p["cy"]), prec, H, W, C.c_void_p(self._d_img.ptr)))
        inter = np.zeros(len(self._colors), np.int64); uni = np.zeros(len(self._colors), np.int64)
        _lib.check(lib.pb3d_partwise_iou_dev(ctx, C.c_void_p(self._d_img.ptr), C.c_void_p(self._d_seg.ptr), H * W, _lib.p_u8(self._colors),
                                             len(self._colors), inter.ctypes.data_as(_lib.i64p), uni.ctypes.data_as(_lib.i64p)))
        per = [(i / u) if u > 0 else 0.0 for i, u in zip(inter, uni)]
        return -np.mean(per)

    # struct pb3d_camera of include/pb3d.h
    _CAM = np.dtype([("R", np.float64, 9), ("cam", np.float64, 3), ("f", np.float64), ("cx", np.float64), ("cy", np.float64),
                     ("prec", np.int32, 4)], align=True)

    def evaluate_batch(self, params):
        """[obj(p) for p in params] with ONE projection launch, one IoU launch and one counter download for the whole list
        (the random / coordinate stages of the aligner evaluate dozens of cameras around the current one).  Exactly the values
        of the one-at-a-time path: the same point kernel per camera, the same integer counts, the same float64 mean."""
        from .camera_geometry import look_at_rotation_batch
        from .projection_utils import _promotes_to_f64
        params = list(params)
        K = len(params)
        if K == 0:
            return []
        for p in params:
            if (int(p.get("H", self.H)), int(p.get("W", self.W))) != (self.H, self.W):
                raise ValueError("operands could not be broadcast together: projection and part image differ in size")
        eyes = [np.asarray(p["cam_pos"]) for p in params]; tgts = [np.asarray(p["target"]) for p in params]
        dts = {a.dtype for a in eyes} | {a.dtype for a in tgts}
        cams = np.zeros(K, self._CAM)
        if len(dts) == 1 and all(a.shape == (3,) for a in eyes) and all(a.shape == (3,) for a in tgts):
            E = np.stack(eyes); T = np.stack(tgts)
            cams["R"] = look_at_rotation_batch(E, T).reshape(K, 9)
            cams["cam"] = E
            t0 = int(np.result_type(self._pts_dtype, E.dtype) == np.float64)     # R has the cameras' dtype (float32 `up` is the narrowest operand)
            t0s = [t0] * K
        else:                                                                    # mixed dtypes: the per-camera NumPy route decides everything
            from .projection_utils import camera_args
            t0s = []
            for k, p in enumerate(params):
                _, _, R, cam, prec = camera_args(np.zeros((1, 3), self._pts_dtype), p["cam_pos"], p["target"], p["f"], p["cx"], p["cy"])
                cams["R"][k] = R.reshape(9); cams["cam"][k] = cam; t0s.append(int(prec[0]))
        for k, p in enumerate(params):
            tm = int(t0s[k] or _promotes_to_f64(p["f"]))
            cams["prec"][k] = (t0s[k], tm, int(tm or _promotes_to_f64(p["cx"])), int(tm or _promotes_to_f64(p["cy"])))
            cams["f"][k] = float(p["f"]); cams["cx"][k] = float(p["cx"]); cams["cy"][k] = float(p["cy"])
        P = len(self._colors)
        inter = np.zeros((K, P), np.int64); uni = np.zeros((K, P), np.int64)
        _lib.check(_lib.load().pb3d_project_iou_batch_dev(
            _lib.ctx(), None if not self.n else C.c_void_p(self._d_pts.ptr), self._pf64, None if not self.n else C.c_void_p(self._d_cols.ptr),
            self.n, cams.ctypes.data_as(C.c_void_p), K, self.H, self.W, C.c_void_p(self._d_seg.ptr), _lib.p_u8(self._colors), P,
            inter.ctypes.data_as(_lib.i64p), uni.ctypes.data_as(_lib.i64p)))
        self.last_counts = (inter, uni)
        with np.errstate(divide="ignore", invalid="ignore"):
            per = np.where(uni > 0, inter / uni, 0.0)
        if 0 < P < 8:
            # fewer than 8 addends: NumPy's reduction is the plain left-to-right sum whichever axis order its iterator
            # picks, so the row-wise mean of the matrix has the bits of np.mean(list) per camera
            return list(-(per.mean(axis=1)))
        return [-np.mean(row) for row in per] if P else [-np.mean([]) for _ in range(K)]

    def projection(self):
        """the image of the last evaluation (H,W,3)"""
        return self._d_img.download((self.H, self.W, 3))

    def close(self):
        for b in (self._d_pts, self._d_cols, self._d_seg, self._d_img):
            if b is not None:
                b.free()


def projection_iou_by_part(voxel_grid, part_colors, image, cam_params):
    """The numbers behind visualize_voxel_projection_iou (reference utils/camera_estimation.py:381-403, :437-452): for every
    part its points are extracted, projected with the camera and compared with the part's pixels of `image`; the
    combined binary IoU compares the union of the projections with every non-background pixel.
    Returns ({part: IoU}, combined_binary_IoU).  The grid is uploaded once and all parts are processed on the device."""
    from . import device as dev
    from .projection_utils import camera_args
    grid = _lib.as_u8(voxel_grid, "voxel_grid")
    img = _lib.as_u8(image, "image")
    H, W = img.shape[:2]
    A0, A1, A2 = grid.shape[:3]
    lib, ctx = _lib.load(), _lib.ctx()
    d_grid = dev.from_numpy(grid); d_img = dev.from_numpy(img); d_proj = dev.DeviceBuffer(H * W * 3)
    bufs = [d_grid, d_img, d_proj]
    per = {}
    union_prj = np.zeros((H, W), bool)
    try:
        _, _, R, cam, prec = camera_args(np.zeros((1, 3), np.float32), cam_params["cam_pos"], cam_params["target"], cam_params["f"],
                                         cam_params["cx"], cam_params["cy"])
        for part, color in part_colors.items():
            c = np.asarray(color).reshape(-1)
            if c.size != 3 or np.any(c < 0) or np.any(c > 255):
                continue
            c8 = np.ascontiguousarray(c.astype(np.uint8))
            n = C.c_int64(0)
            _lib.check(lib.pb3d_points_count_dev(ctx, C.c_void_p(d_grid.ptr), A0, A1, A2, 3, _lib.p_u8(c8), 1, 1, C.byref(n)))
            if n.value == 0:
                continue
            d_pts = dev.DeviceBuffer(n.value * 12); d_pc = dev.DeviceBuffer(n.value * 3)
            try:
                _lib.check(lib.pb3d_points_fill_dev(ctx, C.c_void_p(d_grid.ptr), A0, A1, A2, 3, _lib.p_u8(c8), 1, 1, n.value,
                                                    C.c_void_p(d_pts.ptr), C.c_void_p(d_pc.ptr)))
                _lib.check(lib.pb3d_project_dev(ctx, C.c_void_p(d_pts.ptr), 0, C.c_void_p(d_pc.ptr), n.value, _lib.p_dbl(R), _lib.p_dbl(cam),
                                                float(cam_params["f"]), float(cam_params["cx"]), float(cam_params["cy"]), prec, H, W,
                                                C.c_void_p(d_proj.ptr)))
                inter = np.zeros(1, np.int64); uni = np.zeros(1, np.int64)
                _lib.check(lib.pb3d_partwise_iou_dev(ctx, C.c_void_p(d_proj.ptr), C.c_void_p(d_img.ptr), H * W, _lib.p_u8(c8), 1,
                                                     inter.ctypes.data_as(_lib.i64p), uni.ctypes.data_as(_lib.i64p)))
                per[part] = (inter[0] / uni[0]) if uni[0] > 0 else 0.0
                union_prj |= np.all(d_proj.download((H, W, 3)) == c8, axis=-1)
            finally:
                d_pts.free(); d_pc.free()
        bg = np.array(part_colors.get("background", (0, 0, 0)), dtype=np.uint8)
        gt = np.any(img != bg, axis=-1)
        u = np.logical_or(gt, union_prj).sum()
        return per, ((np.logical_and(gt, union_prj).sum() / u) if u > 0 else 0.0)
    finally:
        for b in bufs:
            b.free()


# =====================================================================================================
# N4: the search loops of launch_smart_aligner as drivers over CameraObjective.evaluate_batch
# (reference utils/camera_estimation.py:606-650 random, :652-686 coordinate descent, :688-726 Powell).
# A parameter set is the dict get_params() builds there (:528-542): cam_pos / target float64 arrays, f / cx / cy floats.
# =====================================================================================================
_KEYS = ("cam_x", "cam_y", "cam_z", "target_x", "target_y", "target_z", "f", "cx", "cy")        # the slider order (:505-518)


def _snap(p):
    q = dict(p)
    q["cam_pos"] = np.array(p["cam_pos"], copy=True); q["target"] = np.array(p["target"], copy=True)
    return q


def random_search(objective, base, steps, rng=None, lock_xy_equal=False):
    """Random Search button (:606-650): `steps` trials around `base` (never around the best so far), uniform in +-(50, 50, 100) for the
    camera and the target, +-50 for f, +-20 for cx / cy, drawn from `rng` (default: the global np.random, as upstream) in upstream's
    order; the best of base and trials by strict `>`.  ALL trials are ONE projection + IoU launch.  Returns (best_params, best_iou)."""
    rng = np.random if rng is None else rng
    step_cam = np.array([50, 50, 100]); step_tgt = np.array([50, 50, 100])
    trials = []
    for _ in range(int(steps)):
        t = dict(base)
        t["cam_pos"] = base["cam_pos"] + rng.uniform(-1, 1, 3) * step_cam
        t["target"] = base["target"] + rng.uniform(-1, 1, 3) * step_tgt
        t["f"] = base["f"] + rng.uniform(-1, 1) * 50
        t["cx"] = base["cx"] + rng.uniform(-1, 1) * 20
        t["cy"] = base["cy"] + rng.uniform(-1, 1) * 20
        if lock_xy_equal:
            t["cam_pos"][:2] = t["target"][:2]
        trials.append(t)
    vals = objective.evaluate_batch([base] + trials)
    best_iou, best_p = -vals[0], dict(base)
    for t, v in zip(trials, vals[1:]):
        if -v > best_iou:
            best_iou, best_p = -v, dict(t)
    return best_p, best_iou


def coordinate_descent(objective, base, rounds, lock_xy_equal=False):
    """Coordinate Descent button (:652-686): per round the +-20 trials of the nine parameters in slider order, the FIRST improvement
    (strict `>`) becomes the new best and ends the round.  Upstream's trial dicts are shallow copies, so the camera / target ARRAYS
    are shared between the best set and every trial and are stepped in place: the -20 trial of an array entry really moves it, the
    +20 trial moves it back (to fl(fl(x - 20) + 20): the value upstream continues with), and only f / cx / cy are ever tried at
    +20.  That sequence does not depend on the IoUs until an improvement ends it, so a round's trials are laid out first and
    evaluated in ONE launch.  Returns (best_params, best_iou)."""
    best_p = _snap(base)
    best_iou = -objective.evaluate_batch([best_p])[0]
    for _ in range(int(rounds)):
        cam = best_p["cam_pos"].copy(); tgt = best_p["target"].copy()          # the shared arrays as the round steps them
        trials = []
        for k in _KEYS:
            for delta in (-20, 20):
                t = dict(best_p)
                if k.startswith("cam_") and not lock_xy_equal:
                    cam["xyz".index(k[-1])] += delta
                elif k.startswith("target_"):
                    tgt["xyz".index(k[-1])] += delta
                    if lock_xy_equal and k in ("target_x", "target_y"):
                        cam["xyz".index(k[-1])] += delta
                elif k in ("f", "cx", "cy"):
                    t[k] = best_p[k] + delta
                else:
                    continue
                t["cam_pos"] = cam.copy(); t["target"] = tgt.copy()
                trials.append(t)
        vals = objective.evaluate_batch(trials)
        for t, v in zip(trials, vals):
            if -v > best_iou:
                best_iou, best_p = -v, t
                break
        else:
            # no improvement: the arrays keep whatever the in-place steps left in them (the scalars are untouched)
            best_p = dict(best_p); best_p["cam_pos"] = cam; best_p["target"] = tgt
    return best_p, best_iou


def powell_search(objective, base, maxiter, minimize, lock_xy_equal=False):
    """Powell button (:688-726) with the caller's minimiser (upstream: `minimize(obj, x0, method='Powell', options=...)`): the
    objective is CameraObjective.__call__ on from_vector(x) (:586-595).  Sequential by nature -- one camera per evaluation.
    Returns (params, iou)."""
    base = _snap(base)
    if lock_xy_equal:
        x0 = np.array([base["cam_pos"][2], base["target"][2], base["f"], base["cx"], base["cy"]])
        tx, ty = base["target"][0], base["target"][1]
        from_vec = lambda x: {"cam_pos": np.array([tx, ty, x[0]]), "target": np.array([tx, ty, x[1]]), "f": x[2], "cx": x[3], "cy": x[4]}
    else:
        x0 = np.concatenate([base["cam_pos"], base["target"], [base["f"], base["cx"], base["cy"]]])
        from_vec = lambda x: {"cam_pos": x[:3], "target": x[3:6], "f": x[6], "cx": x[7], "cy": x[8]}
    res = minimize(lambda x: objective(from_vec(x)), x0, method="Powell",
                   options={"maxiter": int(maxiter), "maxfev": int(maxiter) * 10, "xtol": 1e-3, "ftol": 1e-3, "disp": False})
    p = from_vec(res.x)
    return p, -objective(p)
