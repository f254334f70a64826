"""The 1-byte LABEL form of semantic grids and masks (row N3, device half).

A semantic grid holds only black and the colours of the part palette (reference utils/config.py:29-43; the masks that paint
it come from reference utils/mask_utils.py:14-87), so one byte per voxel says everything: label 0 <-> (0,0,0), label k <->
palette[k-1].  The carve path runs on label volumes at a third of the traffic, a multi-GPU run reassembles labels (1 B/voxel
over xGMI instead of 3), and `label_to_rgb` of any result equals, byte for byte, what the RGB entry points return.

    pal = Palette.from_part_colors(pb3d.PART_COLORS)            # name -> label, (n,3) colours
    lab_mask = pal.mask_to_labels(semantic_mask)                # (H,W) uint8
    grid = global_carve_labels(binary_mask, lab_mask)           # (w,h,w) uint8 labels
    grid = part_carve_labels(grid, lab_mask, group_jobs, pal)   # same jobs as part_carve
    rgb = label_to_rgb(grid, pal)                               # == part_carve(global_carve(binary, semantic), semantic, group_jobs)
"""
import ctypes as C

import numpy as np

from . import _hostmem, _lib

__all__ = ["Palette", "rgb_to_label", "label_to_rgb", "global_carve_labels", "part_carve_labels", "left_right_guided_carve_labels",
           "extrude_from_surface_labels", "recolor_backward_components_labels", "partwise_carve_labels", "get_voxel_points_by_parts_labels",
           "voxel_grid_to_points_labels"]


class Palette:
    """An ordered set of distinct non-black colours; label k is colours[k-1], label 0 is black."""

    def __init__(self, colors, names=None):
        c = np.asarray(colors)
        if c.ndim != 2 or c.shape[1] != 3 or np.any(c < 0) or np.any(c > 255):
            raise ValueError("palette colours must be (n,3) with values in 0..255")
        self.colors = np.ascontiguousarray(c.astype(np.uint8))
        if len(self.colors) > 254:
            raise ValueError("at most 254 colours")
        keys = {tuple(int(v) for v in row) for row in self.colors}
        if len(keys) != len(self.colors) or (0, 0, 0) in keys:
            raise ValueError("palette colours must be distinct and non-black (black is label 0)")
        self.names = list(names) if names is not None else [str(k + 1) for k in range(len(self.colors))]
        if len(self.names) != len(self.colors):
            raise ValueError("one name per colour")

    @classmethod
    def from_part_colors(cls, part_colors):
        names = list(part_colors.keys())
        return cls([part_colors[n] for n in names], names)

    def __len__(self):
        return len(self.colors)

    def label_of(self, name):
        return self.names.index(name) + 1

    def table(self):
        """(n+1, 3) uint8: row k is the colour of label k"""
        return np.concatenate([np.zeros((1, 3), np.uint8), self.colors])

    def mask_to_labels(self, mask_hw3):
        """(H,W,3) palette image -> (H,W) labels (same kernel as the volumes; a 2-D image is a small volume)."""
        m = _lib.as_u8(mask_hw3, "mask")
        if m.ndim != 3 or m.shape[2] != 3:
            raise ValueError("mask must be (H,W,3)")
        return rgb_to_label(m, self)


def _pal(palette):
    return palette if isinstance(palette, Palette) else Palette(palette)


def rgb_to_label(grid_rgb, palette):
    """(..., 3) uint8 colours -> (...) uint8 labels; ValueError if a colour is neither black nor in the palette."""
    from . import device as dev
    pal = _pal(palette)
    g = _lib.as_u8(grid_rgb, "grid")
    if g.ndim < 1 or g.shape[-1] != 3:
        raise ValueError("grid must end in an RGB axis")
    nvox = g.size // 3
    out = _hostmem.empty(g.shape[:-1], np.uint8)
    if nvox == 0:
        return out
    d_in = dev.from_numpy(g); d_out = dev.DeviceBuffer(nvox)
    try:
        _lib.check(_lib.load().pb3d_rgb_to_label_dev(_lib.ctx(), C.c_void_p(d_in.ptr), nvox, _lib.p_u8(pal.colors), len(pal), C.c_void_p(d_out.ptr)))
        return d_out.download(g.shape[:-1])
    finally:
        d_in.free(); d_out.free()


def label_to_rgb(labels, palette):
    """(...) uint8 labels -> (..., 3) uint8 colours."""
    from . import device as dev
    pal = _pal(palette)
    lab = _lib.as_u8(labels, "labels")
    if lab.size == 0:
        return np.zeros(lab.shape + (3,), np.uint8)
    d_in = dev.from_numpy(lab); d_out = dev.DeviceBuffer(lab.size * 3)
    try:
        _lib.check(_lib.load().pb3d_label_to_rgb_dev(_lib.ctx(), C.c_void_p(d_in.ptr), lab.size, _lib.p_u8(pal.colors), len(pal), C.c_void_p(d_out.ptr)))
        return d_out.download(lab.shape + (3,))
    finally:
        d_in.free(); d_out.free()


def global_carve_labels(binary_mask, label_mask, angle_interval=90):
    """global_carve (reference utils/voxel_carving_utils.py:269-298) on labels: (w,h,w) uint8 whose expansion with the
    palette is global_carve(binary_mask, palette_image_of(label_mask), angle_interval)."""
    from . import device as dev
    b = np.asarray(binary_mask)
    if b.ndim != 2:
        raise ValueError("binary_mask must be (h,w)")
    h, w = b.shape
    lm = _lib.as_u8(label_mask, "label_mask")
    if lm.shape != (h, w):
        raise ValueError(f"label mask shape {lm.shape} does not match (h,w)=({h},{w})")
    if isinstance(angle_interval, (bool, np.bool_)) or not isinstance(angle_interval, (int, np.integer)) or angle_interval <= 0:
        raise ValueError("angle_interval must be a positive integer")
    d_b = dev.from_numpy(_lib.truth_u8(b)); d_l = dev.from_numpy(lm); d_out = dev.DeviceBuffer(max(1, w * h * w))
    try:
        _lib.check(_lib.load().pb3d_global_carve_label_dev(_lib.ctx(), C.c_void_p(d_b.ptr), C.c_void_p(d_l.ptr), h, w, int(min(angle_interval, 91)),
                                                           C.c_void_p(d_out.ptr)))
        return d_out.download((w, h, w))
    finally:
        for d in (d_b, d_l, d_out):
            d.free()


def part_carve_labels(label_grid, label_mask, group_jobs, palette):
    """part_carve (reference utils/voxel_carving_utils.py:139-160) on labels.  group_jobs as upstream: [(part names, angle)];
    the names are looked up in the palette."""
    from . import device as dev
    from .voxel_carving_utils import _mask_to_wh
    pal = _pal(palette)
    g = _lib.as_u8(label_grid, "label_grid")
    if g.ndim != 3:
        raise ValueError("label_grid must be (W,H,D)")
    W, H, D = g.shape
    lm = _lib.as_u8(label_mask, "label_mask")
    nj = len(group_jobs)
    msub = np.zeros((max(nj, 1), W, H), np.uint8); mcarve = np.zeros((max(nj, 1), W, H), np.uint8)
    angles = (C.c_int * max(nj, 1))(); skip = (C.c_int * max(nj, 1))()
    for j, (names, angle) in enumerate(group_jobs):
        sel = np.isin(lm, [pal.label_of(n) for n in names])
        skip[j] = 0 if sel.any() else 1
        if angle <= 0 and not skip[j]:
            raise ValueError("job angles must be positive")
        angles[j] = min(int(angle), 91)
        m = sel.T.astype(np.uint8)
        if m.shape != (W, H):
            raise ValueError(f"operands could not be broadcast together: mask {m.shape} vs grid ({W},{H})")
        msub[j] = m
        mcarve[j] = _mask_to_wh(m, W, H)
    if g.size == 0:
        return np.zeros_like(g)
    d_g = dev.from_numpy(g); d_ms = dev.from_numpy(msub); d_mc = dev.from_numpy(mcarve); d_out = dev.DeviceBuffer(g.size)
    try:
        _lib.check(_lib.load().pb3d_part_carve_label_dev(_lib.ctx(), C.c_void_p(d_g.ptr), W, H, D, C.c_void_p(d_ms.ptr), C.c_void_p(d_mc.ptr), angles,
                                                         skip, nj, C.c_void_p(d_out.ptr)))
        return d_out.download(g.shape)
    finally:
        for d in (d_g, d_ms, d_mc, d_out):
            d.free()


# ---- the rest of the notebook-1 chain and the point extraction on label volumes (row N3: "a 1-byte label volume end to end") --------
def _part_carve_labels_dev(d_g, shape3, label_mask, group_jobs, pal, d_out):
    from . import device as dev
    from .voxel_carving_utils import _mask_to_wh
    W, H, D = shape3
    lm = _lib.as_u8(label_mask, "label_mask")
    nj = len(group_jobs)
    msub = np.zeros((max(nj, 1), W, H), np.uint8); mcarve = np.zeros((max(nj, 1), W, H), np.uint8)
    angles = (C.c_int * max(nj, 1))(); skip = (C.c_int * max(nj, 1))()
    for j, (names, angle) in enumerate(group_jobs):
        sel = np.isin(lm, [pal.label_of(n) for n in names])
        skip[j] = 0 if sel.any() else 1
        if angle <= 0 and not skip[j]:
            raise ValueError("job angles must be positive")
        angles[j] = min(int(angle), 91)
        m = sel.T.astype(np.uint8)
        if m.shape != (W, H):
            raise ValueError(f"operands could not be broadcast together: mask {m.shape} vs grid ({W},{H})")
        msub[j] = m
        mcarve[j] = _mask_to_wh(m, W, H)
    d_ms = dev.from_numpy(msub); d_mc = dev.from_numpy(mcarve)
    try:
        _lib.check(_lib.load().pb3d_part_carve_label_dev(_lib.ctx(), C.c_void_p(d_g.ptr), W, H, D, C.c_void_p(d_ms.ptr), C.c_void_p(d_mc.ptr), angles,
                                                         skip, nj, C.c_void_p(d_out.ptr)))
        dev.sync()
    finally:
        d_ms.free(); d_mc.free()


def left_right_guided_carve_labels(label_grid, label_mask, target_label, angle=60, log_color=None):
    """left_right_guided_carve (reference utils/voxel_carving_utils.py:163-210) on a (W,H,D) label volume: the components of
    `target_label`, each carved against label_mask == target_label.  Prints upstream's log (log_color: what to print in the
    "[colour]" place, default the label)."""
    from . import device as dev
    from .voxel_carving_utils import _lrgc_dev
    g = _lib.as_u8(label_grid, "label_grid")
    if g.ndim != 3:
        raise ValueError("label_grid must be (W,H,D)")
    lm = _lib.as_u8(label_mask, "label_mask")
    mask2d = lm == int(target_label)
    shown = target_label if log_color is None else log_color
    if not np.any(mask2d):
        print(f"[SKIP] No mask for color {shown}")
        return g.copy()
    d_g = dev.from_numpy(g)
    try:
        d_r = _lrgc_dev(d_g, g.shape, mask2d, shown, angle, label=int(target_label))
        return d_r.download(g.shape)
    finally:
        d_g.free()


def extrude_from_surface_labels(label_grid, mask_2d, axis, direction="+", depth=5, fill_label=None):
    """extrude_from_surface (reference :213-248) on a label volume; fill_label None clears."""
    from . import device as dev
    from .voxel_carving_utils import _extrude_args, _extrude_dev
    g = _lib.as_u8(label_grid, "label_grid")
    if g.ndim != 3:
        raise ValueError("label_grid must be (W,H,D)")
    if axis not in (0, 2) or int(depth) <= 0:
        return g.copy()
    vt, vw = _extrude_args(g.shape, mask_2d, axis, direction)
    d_in = dev.from_numpy(g); d_out = dev.DeviceBuffer(g.size)
    try:
        _extrude_dev(d_in, d_out, g.shape, vt, vw, axis, direction, depth, fill_label, label=True)
        return d_out.download(g.shape)
    finally:
        d_in.free(); d_out.free()


def recolor_backward_components_labels(label_grid, label, new_label, k=4, sort_axis=2):
    """recolor_backward_components (reference :252-266) on a label volume (any axis order; returns a C-contiguous copy)."""
    from . import device as dev
    from .voxel_carving_utils import _recolor_dev
    g = np.ascontiguousarray(_lib.as_u8(label_grid, "label_grid"))
    if g.ndim != 3:
        raise ValueError("label_grid must be 3-D")
    if g.size == 0:
        return g.copy()
    d_g = dev.from_numpy(g)
    try:
        _recolor_dev(d_g, g.shape, int(label), int(new_label), k, sort_axis, label=True)
        return d_g.download(g.shape)
    finally:
        d_g.free()


def partwise_carve_labels(label_grid, label_mask_exterior, label_mask_full, palette, group_jobs, part_symmetry, extrusion_depths,
                          recolor_back_minarets=True):
    """partwise_carve (reference utils/voxel_carving_utils.py:302-400) on a (W,H,D) label volume, resident from the first stage to
    the last at ONE byte per voxel.  Returns (D,H,W) labels (transposed + flipped, as upstream) when the back-minaret recolouring
    runs, else (W,H,D); label_to_rgb of it is partwise_carve of the RGB grid, byte for byte (the printed log names the colours)."""
    from . import device as dev
    from .voxel_carving_utils import _extrude_args, _extrude_dev, _GuidedParts, _recolor_dev, _recolor_queue
    pal = _pal(palette)
    g = _lib.as_u8(label_grid, "label_grid")
    if g.ndim != 3:
        raise ValueError("label_grid must be (W,H,D)")
    W, H, D = g.shape
    lme = _lib.as_u8(label_mask_exterior, "label_mask_exterior"); lmf = _lib.as_u8(label_mask_full, "label_mask_full")
    d_in = dev.from_numpy(g); d_a = dev.DeviceBuffer(g.size)
    live = [d_in, d_a]
    guided = None
    try:
        _part_carve_labels_dev(d_in, (W, H, D), lme, group_jobs, pal, d_a)
        # all part labels in ONE labelling, their component loops queued behind it (voxel_carving_utils._GuidedParts)
        d_lab = dev.DeviceBuffer(W * H * D * 4)
        live.append(d_lab)
        parts = []
        for part, angle in part_symmetry.items():
            lab = pal.label_of(part)
            parts.append((pal.colors[lab - 1], lme == lab, angle, int(lab)))
        guided = _GuidedParts(d_a, (W, H, D), parts, d_lab, channels=1)
        d_b = guided.run()
        if d_b is not d_a:
            live.append(d_b)
            d_a.free(); live.remove(d_a)
            d_a = d_b
        for part, depth in extrusion_depths.items():
            if int(depth) <= 0:
                continue
            lab = pal.label_of(part)
            mask = lmf == lab
            for axis in (2, 0):
                vt, vw = _extrude_args((W, H, D), mask, axis, "+")
                d_v = dev.from_numpy_async(vt)
                try:
                    for direction in ("+", "-"):
                        _extrude_dev(d_a, d_a, (W, H, D), vt, vw, axis, direction, depth, lab, d_valid=d_v, label=True)
                finally:
                    d_v.free()
        if recolor_back_minarets:
            _lib.check(_lib.load().pb3d_orient_label_dev(_lib.ctx(), C.c_void_p(d_a.ptr), W, H, D, C.c_void_p(d_in.ptr)))
            fm, bm = pal.label_of("front_minarets"), pal.label_of("back_minarets")
            queued = _recolor_queue(d_in, (D, H, W), fm, bm, 2, 0, d_lab, guided.status_ptr(), label=True)
            status = guided.finish(with_status=queued)
            if queued and status[1]:
                _recolor_dev(d_in, (D, H, W), fm, bm, 2, 0, label=True)
            return d_in.download((D, H, W))
        guided.finish()
        return d_a.download(g.shape)
    finally:
        if guided is not None:
            guided.free()
        for b in live:
            b.free()


def _points_labels(label_grid, labels_sel, stride, pal):
    from . import device as dev
    g = _lib.as_u8(label_grid, "label_grid")
    if g.ndim != 3:
        raise ValueError("label_grid must be 3-D")
    A0, A1, A2 = g.shape
    sel = np.ascontiguousarray(np.asarray(labels_sel, np.uint8))
    if g.size == 0 or sel.size == 0:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)
    lib, ctx = _lib.load(), _lib.ctx()
    d_g = dev.from_numpy(g)
    n = C.c_int64(0)
    try:
        _lib.check(lib.pb3d_points_count_dev(ctx, C.c_void_p(d_g.ptr), A0, A1, A2, 1, _lib.p_u8(sel) if sel.size else None, int(sel.size), int(stride), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)
        d_p = dev.DeviceBuffer(n.value * 12); d_l = dev.DeviceBuffer(n.value); d_c = dev.DeviceBuffer(n.value * 3)
        try:
            _lib.check(lib.pb3d_points_fill_dev(ctx, C.c_void_p(d_g.ptr), A0, A1, A2, 1, _lib.p_u8(sel) if sel.size else None, int(sel.size), int(stride),
                                                n.value, C.c_void_p(d_p.ptr), C.c_void_p(d_l.ptr)))
            # the points' colours: their labels expanded with the palette, on the device
            _lib.check(lib.pb3d_label_to_rgb_dev(ctx, C.c_void_p(d_l.ptr), n.value, _lib.p_u8(pal.colors), len(pal), C.c_void_p(d_c.ptr)))
            return d_p.download((n.value, 3), np.float32), d_c.download((n.value, 3))
        finally:
            d_p.free(); d_l.free(); d_c.free()
    finally:
        d_g.free()


def get_voxel_points_by_parts_labels(label_grid, palette, part_names):
    """get_voxel_points_by_parts (reference utils/voxel_utils.py:7-21) on a label volume: the count pass reads 1 B/voxel instead of
    3.  Returns (pts (N,3) float32 in (a2,a1,a0) order, colours (N,3) uint8 from the palette) -- what the RGB entry returns for the
    expanded grid."""
    pal = _pal(palette)
    return _points_labels(label_grid, [pal.label_of(n) for n in part_names], 1, pal)


def voxel_grid_to_points_labels(label_grid, palette, stride=2):
    """voxel_grid_to_points (reference utils/voxel_utils.py:35-51) of the expanded grid: occupied = label != 0, every stride-th voxel,
    points scaled by the stride.  Returns (pts, colours, shape) like upstream."""
    pal = _pal(palette)
    g = _lib.as_u8(label_grid, "label_grid")
    pts, cols = _points_labels_occ(g, stride, pal)
    W, H, D = g.shape
    return pts, cols, (H, W, D)          # (upstream unpacks W, H, D = shape[:3] and returns them in this order, :37 / :51)


def _points_labels_occ(g, stride, pal):
    from . import device as dev
    A0, A1, A2 = g.shape
    if g.size == 0:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)
    lib, ctx = _lib.load(), _lib.ctx()
    d_g = dev.from_numpy(g)
    n = C.c_int64(0)
    try:
        _lib.check(lib.pb3d_points_count_dev(ctx, C.c_void_p(d_g.ptr), A0, A1, A2, 1, None, 0, int(stride), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)
        d_p = dev.DeviceBuffer(n.value * 12); d_l = dev.DeviceBuffer(n.value); d_c = dev.DeviceBuffer(n.value * 3)
        try:
            _lib.check(lib.pb3d_points_fill_dev(ctx, C.c_void_p(d_g.ptr), A0, A1, A2, 1, None, 0, int(stride), n.value, C.c_void_p(d_p.ptr),
                                                C.c_void_p(d_l.ptr)))
            _lib.check(lib.pb3d_label_to_rgb_dev(ctx, C.c_void_p(d_l.ptr), n.value, _lib.p_u8(pal.colors), len(pal), C.c_void_p(d_c.ptr)))
            return d_p.download((n.value, 3), np.float32), d_c.download((n.value, 3))
        finally:
            d_p.free(); d_l.free(); d_c.free()
    finally:
        d_g.free()
