"""Slab sharding of the voxel grid across the GPUs of one node and its reassembly.

The carve path is independent per (x,y) column, so the grid is cut into slabs along its
slowest axis (axis 0: X in the (W,H,D) working orientation; in the stored (D,H,W,3)
artefacts, which are the transposed + flipped grid, working X is stored axis 2).  Slabs are contiguous in memory, every rank runs the same kernels on
its slab with no communication, and ONE RCCL all-gather reassembles the volume.  One
process per GPU; the unique id travels over whatever control plane the launcher provides
(bench.py uses pb3d.rendezvous.ControlPlane, a loopback socket star, for that and for barriers).
"""
import ctypes as C

import numpy as np

from . import _lib


def slab_bounds(n_planes, rank, nranks):
    """[x0, x1) of rank's slab: planes are dealt as evenly as possible, the first
    n_planes % nranks ranks get one extra plane."""
    if not (0 <= rank < nranks):
        raise ValueError(f"rank {rank} outside 0..{nranks - 1}")
    base, extra = divmod(int(n_planes), int(nranks))
    x0 = rank * base + min(rank, extra)
    return x0, x0 + base + (1 if rank < extra else 0)


def equal_slabs(n_planes, nranks):
    """Planes per rank when the partition is even (what a single all-gather needs)."""
    if n_planes % nranks:
        raise ValueError(f"{n_planes} planes do not split evenly over {nranks} ranks; an all-gather needs equal slabs")
    return n_planes // nranks


# ---- Y-slab partition (first row) -----------------------------------------------------------------------
# Every carve op of the path is independent per Y-plane: the rotation is about Y, masks broadcast along Z, colours and part
# selection are per (x,y) column.  A rank that holds the (W, H_r, D) sub-volume of the planes y0..y1 and the rows y0..y1
# of the (H,W) images therefore runs the SAME entry points (carve_voxel_grid_with_masks, process_voxel_grid with any angle
# and any number of chained steps, global_carve, part_carve, apply_colored_mask_to_voxel_grid, extrude_from_surface) on
# its slab with no communication at all; the full volume is the concatenation of the slabs along axis 1.  (Connected
# components cross planes and do not shard this way.  part_carve on a grid with W == H inherits the reference's square-mask
# quirk -- _mask_to_wh transposes the part mask a second time -- which a slab with H_r != W does not reproduce: partition
# part_carve by Y on non-square grids, as the reference's own (w, h, w) grids with h != w are.)
def y_slab_bounds(n_rows, rank, nranks):
    """[y0, y1) of rank's share of the H image rows / grid planes (same dealing rule as slab_bounds)."""
    return slab_bounds(n_rows, rank, nranks)


def y_slab_image(image_hw, rank, nranks):
    """Rows y0..y1 of an (H,W) or (H,W,3) image -- the mask this rank passes with its (W, y1-y0, D) sub-volume."""
    a = np.asarray(image_hw)
    y0, y1 = y_slab_bounds(a.shape[0], rank, nranks)
    return np.ascontiguousarray(a[y0:y1])


def y_slab_grid(grid_whd, rank, nranks):
    """Planes y0..y1 (axis 1) of a (W,H,D[,3]) grid as a contiguous sub-volume."""
    g = np.asarray(grid_whd)
    y0, y1 = y_slab_bounds(g.shape[1], rank, nranks)
    return np.ascontiguousarray(g[:, y0:y1])


def assemble_y_slabs(slabs):
    """Full (W,H,D[,3]) volume from the ranks' sub-volumes, in rank order."""
    return np.concatenate([np.asarray(s) for s in slabs], axis=1)


def new_unique_id():
    uid = np.zeros(128, np.uint8)
    _lib.check(_lib.load().pb3d_comm_unique_id(_lib.p_u8(uid)))
    return uid


def comm_init(unique_id, rank, nranks):
    uid = np.ascontiguousarray(unique_id, np.uint8)
    _lib.check(_lib.load().pb3d_comm_init(_lib.ctx(), _lib.p_u8(uid), int(rank), int(nranks)))


def comm_info():
    """(rank, nranks) as the RCCL communicator itself reports them."""
    r, n = C.c_int(-1), C.c_int(-1)
    _lib.check(_lib.load().pb3d_comm_info(_lib.ctx(), C.byref(r), C.byref(n)))
    return r.value, n.value


def allgather(d_send, d_recv, bytes_per_rank):
    """Enqueue the slab all-gather on the context stream (d_send may be the rank's slot of d_recv)."""
    from .device import _ptr
    _lib.check(_lib.load().pb3d_allgather_dev(_lib.ctx(), _ptr(d_send), _ptr(d_recv), int(bytes_per_rank)))


def carve_mask_sharded(d_grid_slab, W, H, D, channels, d_mask_wh, d_out_full):
    """This rank's slab carved into its slot of d_out_full + ONE in-place all-gather (pb3d_carve_mask_sharded_dev); enqueue only."""
    from .device import _ptr
    _lib.check(_lib.load().pb3d_carve_mask_sharded_dev(_lib.ctx(), _ptr(d_grid_slab), int(W), int(H), int(D), int(channels), _ptr(d_mask_wh),
                                                       _ptr(d_out_full)))


def global_carve_sharded(d_bin_hw, d_rgb_hw3, h, w, angle_interval, d_out_full):
    from .device import _ptr
    _lib.check(_lib.load().pb3d_global_carve_sharded_dev(_lib.ctx(), _ptr(d_bin_hw), _ptr(d_rgb_hw3), int(h), int(w), int(angle_interval),
                                                         _ptr(d_out_full)))


def carve_labels_sharded(d_label_slab, W, H, D, d_mask_wh, d_label_full, palette_colors=None, d_rgb_full=None):
    """The compact form of the reassembly: the slab is carved as LABELS (1 B/voxel), ONE all-gather moves a
    third of the bytes of the RGB form, and -- only if d_rgb_full is given -- the reassembled label volume is expanded to RGB
    locally.  palette_colors: (n,3) uint8 (label k <-> palette_colors[k-1])."""
    from .device import _ptr
    pal = np.zeros((0, 3), np.uint8) if palette_colors is None else np.ascontiguousarray(palette_colors, np.uint8)
    _lib.check(_lib.load().pb3d_carve_labels_sharded_dev(_lib.ctx(), _ptr(d_label_slab), int(W), int(H), int(D), _ptr(d_mask_wh), _ptr(d_label_full),
                                                         _lib.p_u8(pal), len(pal), _ptr(d_rgb_full)))


def comm_destroy():
    _lib.check(_lib.load().pb3d_comm_destroy(_lib.ctx()))


def point_shard_bounds(n_points, rank, nranks):
    """[i0, i1) of rank's contiguous share of a point list (same dealing rule as slab_bounds)."""
    return slab_bounds(n_points, rank, nranks)


def project_colored_voxels_sharded(pts_shard, colors_shard, index_base, cam_pos, target, f, cx, cy, H, W, reduce=True):
    """project_colored_voxels (reference utils/projection_utils.py:5-23) with the point list partitioned over the ranks:
    `pts_shard` / `colors_shard` are this rank's contiguous range starting at global index `index_base`.  Each rank
    scatters into a private image of (index, colour) keys; ONE all-reduce(max) merges them and every rank resolves
    the same (H,W,3) image the unsharded call returns.  reduce=False skips the collective (single rank, or a caller
    that merges key images itself) and returns (image, keys)."""
    import ctypes as C
    import numpy as np
    from . import device as dev
    from .projection_utils import camera_args
    p, pf64, R, cam, prec = camera_args(pts_shard, cam_pos, target, f, cx, cy)
    cols = np.ascontiguousarray(np.asarray(colors_shard).astype(np.uint8, copy=False))
    if cols.shape != (len(p), 3):
        raise ValueError("colors must be (N,3)")
    lib, ctx = _lib.load(), _lib.ctx()
    d_p = dev.from_numpy(p) if len(p) else None
    d_c = dev.from_numpy(cols) if len(p) else None
    d_keys = dev.DeviceBuffer(int(H) * int(W) * 8); d_img = dev.DeviceBuffer(int(H) * int(W) * 3)
    try:
        _lib.check(lib.pb3d_project_keys_dev(ctx, None if d_p is None else C.c_void_p(d_p.ptr), pf64, None if d_c is None else C.c_void_p(d_c.ptr),
                                             len(p), int(index_base), _lib.p_dbl(R), _lib.p_dbl(cam), float(f), float(cx), float(cy), prec,
                                             int(H), int(W), C.c_void_p(d_keys.ptr)))
        if reduce:
            _lib.check(lib.pb3d_allreduce_max_u64_dev(ctx, C.c_void_p(d_keys.ptr), int(H) * int(W)))
        _lib.check(lib.pb3d_project_resolve_keys_dev(ctx, C.c_void_p(d_keys.ptr), int(H), int(W), C.c_void_p(d_img.ptr)))
        img = d_img.download((int(H), int(W), 3))
        return img if reduce else (img, d_keys.download((int(H), int(W)), np.uint64))
    finally:
        for b in (d_p, d_c, d_keys, d_img):
            if b is not None:
                b.free()


def resolve_keys(keys):
    """(H,W) uint64 key image -> (H,W,3) uint8 colours (host helper for callers that merged key images themselves)."""
    import numpy as np
    k = np.asarray(keys, np.uint64)
    return np.stack([(k & np.uint64(0xff)), (k >> np.uint64(8)) & np.uint64(0xff), (k >> np.uint64(16)) & np.uint64(0xff)], axis=-1).astype(np.uint8)
