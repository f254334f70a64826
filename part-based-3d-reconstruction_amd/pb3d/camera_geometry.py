"""Look-at rotation and single-point projection (host side).

Mirrors reference utils/camera_geometry.py:3-27.  These are 3-vector operations whose
NumPy evaluation order and dtypes (float32 from the JSON path, float64 from the slider /
optimiser path) decide the bits of R that the projection kernel consumes, so they stay in
NumPy on the host; the per-point work is in csrc/project.hip.
"""
import numpy as np

_UP_Y = np.array([0, 1, 0], dtype=np.float32)


def look_at_rotation(eye, target, up=_UP_Y):
    """Rows (x, y, z) of the camera frame looking from `eye` to `target`."""
    forward = target - eye
    forward /= np.linalg.norm(forward)          # in place, as upstream: mutates a fresh array only
    if np.allclose(np.abs(np.dot(forward, up)), 1.0):
        up = np.array([0, 0, 1], dtype=np.float32)
    right = np.cross(up, forward)
    right /= np.linalg.norm(right)
    true_up = np.cross(forward, right)
    return np.stack([right, true_up, forward], axis=0)


_dot_mode = {}      # np.dtype -> dot_mode of pb3d_look_at_batch that reproduces this host's NumPy, or None


def _calibrate(dtype):
    """Find the rounding of the 3-element dot product inside numpy.linalg.norm on THIS host (it belongs to the BLAS kernel
    NumPy dispatched to, not to NumPy's source) by comparing pb3d_look_at_batch with look_at_rotation on a probe set that
    includes straight-up / straight-down and zero-length views.  None: no mode reproduces NumPy -- callers stay on NumPy."""
    import ctypes as C
    from . import _lib
    rng = np.random.default_rng(20261004)
    K = 192
    eye = (rng.normal(size=(K, 3)) * rng.choice([1.0, 30.0, 1000.0], size=(K, 1))).astype(dtype)
    tgt = (rng.normal(size=(K, 3)) * 40.0).astype(dtype)
    eye[:8, 0] = tgt[:8, 0]; eye[:8, 2] = tgt[:8, 2]          # forward parallel to the default up vector
    eye[8:12] = tgt[8:12]                                       # zero-length forward (NaNs, as upstream)
    with np.errstate(all="ignore"):
        want = np.stack([np.asarray(look_at_rotation(eye[k].copy(), tgt[k].copy()), np.float64) for k in range(K)])
    lib = _lib.load()
    is64 = int(dtype == np.float64)
    got = np.empty((K, 3, 3), np.float64)
    for mode in ((1, 0, 3) if is64 else (4, 0, 1, 2, 3)):
        if lib.pb3d_look_at_batch(eye.ctypes.data_as(C.c_void_p), tgt.ctypes.data_as(C.c_void_p), is64, K, mode, _lib.p_dbl(got)) == 0 \
                and np.array_equal(got, want, equal_nan=True):
            return mode
    return None


def look_at_rotation_batch(eyes, targets):
    """(K,3,3) float64 array whose k-th matrix has exactly the bits of look_at_rotation(eyes[k], targets[k]) cast to float64
    (eyes, targets: (K,3) arrays of one float dtype).  Native when the host's NumPy rounding could be calibrated, else a
    NumPy loop."""
    import ctypes as C
    from . import _lib
    eyes = np.asarray(eyes); targets = np.asarray(targets)
    if eyes.shape != targets.shape or eyes.ndim != 2 or eyes.shape[1] != 3:
        raise ValueError("eyes and targets must both be (K,3)")
    K = len(eyes)
    dt = eyes.dtype
    if dt == targets.dtype and dt in (np.float32, np.float64):
        if dt not in _dot_mode:
            _dot_mode[dt] = _calibrate(dt)
        mode = _dot_mode[dt]
        if mode is not None:
            e = np.ascontiguousarray(eyes); t = np.ascontiguousarray(targets)
            R = np.empty((K, 3, 3), np.float64)
            _lib.check(_lib.load().pb3d_look_at_batch(e.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), int(dt == np.float64), K, mode,
                                                      _lib.p_dbl(R)))
            return R
    with np.errstate(all="ignore"):
        return np.stack([np.asarray(look_at_rotation(np.array(eyes[k]), np.array(targets[k])), np.float64) for k in range(K)]) if K else np.empty((0, 3, 3))


def project(pt3d, cam_pos, target, f, cx, cy):
    """Pinhole projection of one point (keypoint path; scalar, stays on the host)."""
    R = look_at_rotation(cam_pos, target)
    X, Y, Z = (pt3d - cam_pos) @ R.T
    Z = max(Z, 1e-8)
    return np.array([(X / Z) * f + cx, -(Y / Z) * f + cy])
