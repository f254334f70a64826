"""ctypes binding of libpb3d.so (include/pb3d.h) and the per-process default context.

There is no CPU fallback: if the shared library is missing, or no MI355X is visible, every
operation raises.  Nothing here imports torch or anything under oracle/.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PB3D_LIB_PATH") or os.path.join(_HERE, "libpb3d.so")   # override: A/B builds of the same ABI

u8p = C.POINTER(C.c_uint8)
i64 = C.c_int64
i64p = C.POINTER(C.c_int64)
dblp = C.POINTER(C.c_double)
intp = C.POINTER(C.c_int)
vp = C.c_void_p

# name -> argtypes ; every function returns int unless listed in _RESTYPES
_SIGNATURES = {
    "pb3d_version": [],
    "pb3d_last_error": [],
    "pb3d_device_count": [intp],
    "pb3d_create": [C.c_int, C.POINTER(vp)],
    "pb3d_destroy": [vp],
    "pb3d_make_current": [vp],
    "pb3d_device_info": [vp, C.c_char_p, C.c_int, intp, i64p],
    "pb3d_sync": [vp],
    "pb3d_set_tuning": [vp, C.c_char_p, C.c_int],
    "pb3d_dev_alloc": [vp, C.c_size_t, C.POINTER(vp)],
    "pb3d_dev_free": [vp, vp],
    "pb3d_dev_memset": [vp, vp, C.c_int, C.c_size_t],
    "pb3d_h2d": [vp, vp, vp, C.c_size_t],
    "pb3d_d2h": [vp, vp, vp, C.c_size_t],
    "pb3d_h2d_async": [vp, vp, vp, C.c_size_t],
    "pb3d_sync_count": [vp],
    "pb3d_d2d": [vp, vp, vp, C.c_size_t],
    "pb3d_event_create": [vp, C.POINTER(vp)],
    "pb3d_event_record": [vp, vp],
    "pb3d_event_elapsed_ms": [vp, vp, vp, C.POINTER(C.c_float)],
    "pb3d_event_destroy": [vp],
    "pb3d_rotinv": [C.c_int, dblp],
    "pb3d_offset": [dblp, i64p, dblp],
    "pb3d_carve_mask_dev": [vp, vp, i64, i64, i64, C.c_int, vp, vp],
    "pb3d_carve_mask": [vp, u8p, i64, i64, i64, C.c_int, u8p, u8p],
    "pb3d_rotate_carve_dev": [vp, vp, i64, i64, i64, dblp, dblp, vp, vp],
    "pb3d_rotate_carve": [vp, u8p, i64, i64, i64, dblp, dblp, u8p, u8p],
    "pb3d_process_grid_dev": [vp, vp, i64, i64, i64, vp, C.c_int, vp, vp],
    "pb3d_process_grid": [vp, u8p, i64, i64, i64, u8p, C.c_int, u8p],
    "pb3d_dtype_bytes": [C.c_int],
    "pb3d_process_grid_typed_dev": [vp, vp, C.c_int, i64, i64, i64, vp, C.c_int, vp, vp],
    "pb3d_occupancy_dev": [vp, vp, i64, vp],
    "pb3d_occupancy": [vp, u8p, i64, u8p],
    "pb3d_color_apply_dev": [vp, vp, i64, i64, i64, vp, vp],
    "pb3d_color_apply": [vp, u8p, i64, i64, i64, u8p, u8p],
    "pb3d_global_carve_dev": [vp, vp, vp, i64, i64, C.c_int, i64, i64, vp],
    "pb3d_global_carve": [vp, u8p, u8p, i64, i64, C.c_int, u8p],
    "pb3d_part_carve_dev": [vp, vp, i64, i64, i64, vp, vp, intp, intp, C.c_int, vp],
    "pb3d_part_carve": [vp, u8p, i64, i64, i64, u8p, u8p, intp, intp, C.c_int, u8p],
    "pb3d_points_count_dev": [vp, vp, i64, i64, i64, C.c_int, u8p, C.c_int, C.c_int, i64p],
    "pb3d_points_fill_dev": [vp, vp, i64, i64, i64, C.c_int, u8p, C.c_int, C.c_int, i64, vp, vp],
    "pb3d_points_extract_dev": [vp, vp, i64, i64, i64, C.c_int, u8p, C.c_int, i64, vp, vp, i64p],
    "pb3d_points_count": [vp, u8p, i64, i64, i64, C.c_int, u8p, C.c_int, C.c_int, i64p],
    "pb3d_points_fill": [vp, i64, C.POINTER(C.c_float), u8p],
    "pb3d_project_dev": [vp, vp, C.c_int, vp, i64, dblp, dblp, C.c_double, C.c_double, C.c_double, intp, C.c_int, C.c_int, vp],
    "pb3d_project": [vp, vp, C.c_int, u8p, i64, dblp, dblp, C.c_double, C.c_double, C.c_double, intp, C.c_int, C.c_int, u8p],
    "pb3d_partwise_iou_dev": [vp, vp, vp, i64, u8p, C.c_int, i64p, i64p],
    "pb3d_project_iou_batch_dev": [vp, vp, C.c_int, vp, i64, vp, C.c_int, C.c_int, C.c_int, vp, u8p, C.c_int, i64p, i64p],
    "pb3d_look_at_batch": [vp, vp, C.c_int, i64, C.c_int, dblp],
    "pb3d_deform_iou_batch_dev": [vp, vp, i64, dblp, C.c_int, i64, i64, i64, vp, C.c_int, C.c_int, vp, u8p, i64p, i64p, i64p],
    "pb3d_depth_buffer_dev": [vp, vp, C.c_int, i64, dblp, dblp, C.c_double, C.c_double, C.c_double, intp, C.c_int, C.c_int, vp],
    "pb3d_visible_mask_dev": [vp, vp, C.c_int, i64, dblp, dblp, C.c_double, C.c_double, C.c_double, intp, vp, C.c_int, C.c_int,
                              C.c_double, C.c_int, vp],
    "pb3d_partwise_iou": [vp, u8p, u8p, i64, u8p, C.c_int, i64p, i64p],
    "pb3d_deform_count_dev": [vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i64p],
    "pb3d_deform_fill_dev": [vp, i64, vp],
    "pb3d_deform_count": [vp, C.POINTER(C.c_float), i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i64p],
    "pb3d_deform_fill": [vp, i64, i64p],
    "pb3d_deform_paint_dev": [vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i64, i64, i64, u8p, vp],
    "pb3d_scatter_colors_dev": [vp, vp, vp, i64, i64, i64, i64, vp],
    "pb3d_label_color_dev": [vp, vp, i64, i64, i64, u8p, vp, i64p],
    "pb3d_label_color_stats_dev": [vp, vp, i64, i64, i64, u8p, vp, i64p, i64, C.c_int, i64p, i64p, i64p, intp],
    "pb3d_label_colors_stats_dev": [vp, vp, i64, i64, i64, u8p, C.c_int, vp, i64p, i64, C.c_int, i64p, i64p, i64p, intp],
    "pb3d_label_values_stats_dev": [vp, vp, i64, i64, i64, u8p, C.c_int, vp, i64p, i64, C.c_int, i64p, i64p, i64p, intp],
    "pb3d_guided_carve_color_dev": [vp, vp, vp, C.c_int, C.c_int, i64, i64, i64, i64, i64p, u8p, i64p, i64, C.c_int, i64p, intp],
    "pb3d_guided_carve_queue_dev": [vp, vp, vp, C.c_int, C.c_int, i64, i64, i64, i64, i64p, u8p, i64p, i64, C.c_int, vp, intp],
    "pb3d_recolor_backward_dev": [vp, vp, i64, i64, i64, u8p, u8p, C.c_int, C.c_int, C.c_int, vp, vp],
    "pb3d_component_stats_dev": [vp, vp, i64, i64, i64, i64, i64p, i64p, i64p],
    "pb3d_crop_occupancy_dev": [vp, vp, i64, i64, i64, i64p, i64p, vp],
    "pb3d_crop_occupancy_label_dev": [vp, vp, i64, i64, i64, i64p, i64p, vp],
    "pb3d_component_paste_dev": [vp, vp, vp, C.c_int32, vp, i64, i64, i64, i64p, i64p, vp],
    "pb3d_component_paste_label_dev": [vp, vp, vp, C.c_int32, vp, i64, i64, i64, i64p, i64p, vp],
    "pb3d_guided_carve_dev": [vp, vp, vp, i64, i64, i64, i64, i64p, u8p, i64p, i64, C.c_int, i64p, intp],
    "pb3d_guided_carve_label_dev": [vp, vp, vp, i64, i64, i64, i64, i64p, u8p, i64p, i64, C.c_int, i64p, intp],
    "pb3d_label_value_stats_dev": [vp, vp, i64, i64, i64, C.c_uint8, vp, i64p, i64, C.c_int, i64p, i64p, i64p, intp],
    "pb3d_extrude_label_dev": [vp, vp, i64, i64, i64, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp],
    "pb3d_recolor_components_label_dev": [vp, vp, i64, u8p, i64, C.c_uint8, vp],
    "pb3d_recolor_last_labelled_dev": [vp, vp, i64, u8p, i64, u8p, vp, C.c_int],
    "pb3d_orient_label_dev": [vp, vp, i64, i64, i64, vp],
    "pb3d_count_nonzero_dev": [vp, vp, i64, vp],
    "pb3d_recolor_components_dev": [vp, vp, i64, u8p, i64, u8p, vp],
    "pb3d_extrude_dev": [vp, vp, i64, i64, i64, vp, i64, C.c_int, C.c_int, C.c_int, u8p, vp],
    "pb3d_orient_dev": [vp, vp, i64, i64, i64, vp],
    "pb3d_synth_mask16_dev": [vp, i64, vp, vp, vp, vp],
    "pb3d_synth_sem_dev": [vp, i64, i64, i64, i64, C.c_uint64, vp],
    "pb3d_synth_occ_dev": [vp, i64, i64, i64, i64, C.c_uint64, vp],
    "pb3d_synth_palette16": [u8p],
    "pb3d_comm_unique_id": [u8p],
    "pb3d_comm_init": [vp, u8p, C.c_int, C.c_int],
    "pb3d_comm_info": [vp, intp, intp],
    "pb3d_allgather_dev": [vp, vp, vp, C.c_size_t],
    "pb3d_comm_destroy": [vp],
    "pb3d_carve_mask_sharded_dev": [vp, vp, i64, i64, i64, C.c_int, vp, vp],
    "pb3d_global_carve_sharded_dev": [vp, vp, vp, i64, i64, C.c_int, vp],
    "pb3d_carve_labels_sharded_dev": [vp, vp, i64, i64, i64, vp, vp, u8p, C.c_int, vp],
    "pb3d_rgb_to_label_dev": [vp, vp, i64, u8p, C.c_int, vp],
    "pb3d_label_to_rgb_dev": [vp, vp, i64, u8p, C.c_int, vp],
    "pb3d_global_carve_label_dev": [vp, vp, vp, i64, i64, C.c_int, vp],
    "pb3d_part_carve_label_dev": [vp, vp, i64, i64, i64, vp, vp, intp, intp, C.c_int, vp],
    "pb3d_project_keys_dev": [vp, vp, C.c_int, vp, i64, i64, dblp, dblp, C.c_double, C.c_double, C.c_double, intp, C.c_int, C.c_int, vp],
    "pb3d_project_resolve_keys_dev": [vp, vp, C.c_int, C.c_int, vp],
    "pb3d_allreduce_max_u64_dev": [vp, vp, C.c_size_t],
}
_RESTYPES = {"pb3d_dtype_bytes": C.c_size_t, "pb3d_sync_count": C.c_int64, "pb3d_last_error": C.c_char_p, "pb3d_destroy": None, "pb3d_event_destroy": None}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_ctx = None
_lock = threading.Lock()


class Pb3dError(RuntimeError):
    pass


def load():
    """Load libpb3d.so (raises ImportError if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(pb3d has no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().pb3d_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise Pb3dError(f"libpb3d error {rc}: {msg}")


def device_count():
    n = C.c_int(0)
    check(load().pb3d_device_count(C.byref(n)))
    return n.value


def default_device():
    for key in ("PB3D_DEVICE", "LOCAL_RANK"):
        if os.environ.get(key, "").isdigit():
            return int(os.environ[key])
    return 0


def ctx():
    """The process-wide context (one GPU, one HIP stream), created on first use."""
    global _ctx
    with _lock:
        if _ctx is None:
            h = vp()
            check(load().pb3d_create(default_device(), C.byref(h)))
            _ctx = h
    return _ctx


def set_tuning(name, value):
    """Development knob of the process context (include/pb3d.h: pb3d_set_tuning); results never depend on it."""
    check(load().pb3d_set_tuning(ctx(), name.encode(), int(value)))


def reset():
    global _ctx
    with _lock:
        if _ctx is not None:
            load().pb3d_destroy(_ctx)
            _ctx = None


def p_u8(a):
    return a.ctypes.data_as(u8p)


def p_dbl(a):
    return a.ctypes.data_as(dblp)


def as_u8(a, what):
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError(f"{what} must be a uint8 array (got {a.dtype}); the carving path stores grids as uint8")
    return np.ascontiguousarray(a)


def truth_u8(a):
    """uint8 0/1 image of NumPy truthiness (what np.where(mask, ...) keys on)."""
    return np.ascontiguousarray(np.asarray(a) != 0).view(np.uint8)
