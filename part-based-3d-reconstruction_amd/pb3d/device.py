"""Device-resident operation layer: HBM buffers, events and the *_dev entry points.

This is what pipelines that keep the volume in HBM between steps (and bench.py) use; the
NumPy-signature functions in the sibling modules are thin host-staging wrappers of the
same kernels.  All launches go to the context's single HIP stream.
"""
import ctypes as C

import numpy as np

from . import _hostmem, _lib


class DeviceBuffer:
    """A hipMalloc'ed region owned by the process context."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _lib.check(_lib.load().pb3d_dev_alloc(_lib.ctx(), self.nbytes, C.byref(p)))
        self.ptr = p.value

    def at(self, byte_offset):
        if not 0 <= byte_offset <= self.nbytes:
            raise ValueError("offset outside the buffer")
        return C.c_void_p(self.ptr + int(byte_offset))

    def upload(self, array, byte_offset=0):
        a = np.ascontiguousarray(array)
        if byte_offset + a.nbytes > self.nbytes:
            raise ValueError("upload exceeds the buffer")
        _lib.check(_lib.load().pb3d_h2d(_lib.ctx(), self.at(byte_offset), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return self

    def upload_async(self, array, byte_offset=0):
        """upload without waiting: the bytes are staged in the context's pinned ring (pb3d_h2d_async), `array` may be reused at once"""
        a = np.ascontiguousarray(array)
        if byte_offset + a.nbytes > self.nbytes:
            raise ValueError("upload exceeds the buffer")
        _lib.check(_lib.load().pb3d_h2d_async(_lib.ctx(), self.at(byte_offset), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return self

    def download(self, shape, dtype=np.uint8, byte_offset=0):
        out = _hostmem.empty(shape, dtype)
        if byte_offset + out.nbytes > self.nbytes:
            raise ValueError("download exceeds the buffer")
        _lib.check(_lib.load().pb3d_d2h(_lib.ctx(), out.ctypes.data_as(C.c_void_p), self.at(byte_offset), out.nbytes))
        return out

    def zero(self):
        _lib.check(_lib.load().pb3d_dev_memset(_lib.ctx(), C.c_void_p(self.ptr), 0, self.nbytes))

    def free(self):
        if self.ptr:
            _lib.check(_lib.load().pb3d_dev_free(_lib.ctx(), C.c_void_p(self.ptr)))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceGrid:
    """A (W,H,D[,3]) uint8 voxel grid that lives in HBM.  global_carve(..., on_device=True) returns one; part_carve and
    partwise_carve accept one in place of the NumPy grid and then return one too, so the notebook-1 chain runs without a
    single upload or download of the volume; .numpy() brings the result back, .free() releases the memory."""

    def __init__(self, buf, shape):
        self.buf = buf
        self.shape = tuple(int(v) for v in shape)
        self.dtype = np.dtype(np.uint8)

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def numpy(self):
        return self.buf.download(self.shape)

    def free(self):
        self.buf.free()

    def save_npz(self, path):
        """the stage hand-off file of the notebooks (np.savez_compressed with the single key `voxel_grid`); pb3d/formats.py"""
        from .formats import save_voxel_grid
        save_voxel_grid(path, self)

    @classmethod
    def load_npz(cls, path):
        from .formats import load_voxel_grid
        return load_voxel_grid(path, on_device=True)


def from_numpy(array):
    a = np.ascontiguousarray(array)
    return DeviceBuffer(a.nbytes).upload(a)


def from_numpy_async(array):
    """from_numpy without a host wait (small inputs: 2-D masks, descriptors)"""
    a = np.ascontiguousarray(array)
    return DeviceBuffer(a.nbytes).upload_async(a)


def sync_count():
    """how often the host has waited for the context's stream so far"""
    return int(_lib.load().pb3d_sync_count(_lib.ctx()))


class Event:
    def __init__(self):
        p = C.c_void_p()
        _lib.check(_lib.load().pb3d_event_create(_lib.ctx(), C.byref(p)))
        self.h = p

    def record(self):
        _lib.check(_lib.load().pb3d_event_record(_lib.ctx(), self.h))
        return self

    def elapsed_ms_since(self, start):
        ms = C.c_float(0)
        _lib.check(_lib.load().pb3d_event_elapsed_ms(_lib.ctx(), start.h, self.h, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            _lib.load().pb3d_event_destroy(self.h)
        except Exception:
            pass


def sync():
    _lib.check(_lib.load().pb3d_sync(_lib.ctx()))


def device_info():
    name = C.create_string_buffer(128)
    cus = C.c_int(0)
    mem = C.c_int64(0)
    _lib.check(_lib.load().pb3d_device_info(_lib.ctx(), name, 128, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": mem.value}


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return C.c_void_p(x.ptr)
    return x  # already a c_void_p (buffer.at(offset))


# ---- device-resident ops (enqueue only; call sync() or an Event to wait) --------------------------
def carve_mask(d_grid, W, H, D, channels, d_mask_wh, d_out):
    _lib.check(_lib.load().pb3d_carve_mask_dev(_lib.ctx(), _ptr(d_grid), W, H, D, channels, _ptr(d_mask_wh), _ptr(d_out)))


def process_grid(d_occ, W, H, D, d_mask_wh, angle_interval, d_out, d_tmp):
    _lib.check(_lib.load().pb3d_process_grid_dev(_lib.ctx(), _ptr(d_occ), W, H, D, _ptr(d_mask_wh), int(angle_interval),
                                                 _ptr(d_out), _ptr(d_tmp)))


def rotate_carve(d_occ, W, H, D, M, off, d_mask_wh, d_out):
    M = np.ascontiguousarray(M, np.float64); off = np.ascontiguousarray(off, np.float64)
    _lib.check(_lib.load().pb3d_rotate_carve_dev(_lib.ctx(), _ptr(d_occ), W, H, D, _lib.p_dbl(M), _lib.p_dbl(off),
                                                 _ptr(d_mask_wh), _ptr(d_out)))


def occupancy(d_grid_rgb, nvox, d_occ):
    _lib.check(_lib.load().pb3d_occupancy_dev(_lib.ctx(), _ptr(d_grid_rgb), nvox, _ptr(d_occ)))


def color_apply(d_carved, W, H, D, d_rgb_hw3, d_out):
    _lib.check(_lib.load().pb3d_color_apply_dev(_lib.ctx(), _ptr(d_carved), W, H, D, _ptr(d_rgb_hw3), _ptr(d_out)))


def global_carve(d_bin_hw, d_rgb_hw3, h, w, angle_interval, d_out_slab, x0=0, x1=None):
    _lib.check(_lib.load().pb3d_global_carve_dev(_lib.ctx(), _ptr(d_bin_hw), _ptr(d_rgb_hw3), h, w, int(angle_interval), x0,
                                                 w if x1 is None else x1, _ptr(d_out_slab)))


def rgb_to_label(d_rgb, nvox, palette_colors, d_label):
    """synchronous (reads back the unknown-colour flag)"""
    pal = np.ascontiguousarray(palette_colors, np.uint8)
    _lib.check(_lib.load().pb3d_rgb_to_label_dev(_lib.ctx(), _ptr(d_rgb), int(nvox), _lib.p_u8(pal), len(pal), _ptr(d_label)))


def label_to_rgb(d_label, nvox, palette_colors, d_rgb):
    pal = np.ascontiguousarray(palette_colors, np.uint8)
    _lib.check(_lib.load().pb3d_label_to_rgb_dev(_lib.ctx(), _ptr(d_label), int(nvox), _lib.p_u8(pal), len(pal), _ptr(d_rgb)))


def synth_mask16(S, d_label_hw=None, d_binary_hw=None, d_rgb_hw3=None, d_binary_wh=None):
    _lib.check(_lib.load().pb3d_synth_mask16_dev(_lib.ctx(), S, _ptr(d_label_hw), _ptr(d_binary_hw), _ptr(d_rgb_hw3),
                                                 _ptr(d_binary_wh)))


def synth_sem(x0, x1, H, D, seed, d_slab_rgb):
    _lib.check(_lib.load().pb3d_synth_sem_dev(_lib.ctx(), x0, x1, H, D, seed, _ptr(d_slab_rgb)))


def synth_occ(x0, x1, H, D, seed, d_slab):
    _lib.check(_lib.load().pb3d_synth_occ_dev(_lib.ctx(), x0, x1, H, D, seed, _ptr(d_slab)))


def synth_palette16():
    pal = np.zeros((16, 3), np.uint8)
    _lib.check(_lib.load().pb3d_synth_palette16(_lib.p_u8(pal)))
    return pal
