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


def project(pt3d, cam_pos, target, f, cx, cy):
    """Pinhole projection of one point (keypoint path; scalar, stays on the host)."""
    R = look_at_rotation(cam_pos, target)
    X, Y, Z = (pt3d - cam_pos) @ R.T
    Z = max(Z, 1e-8)
    return np.array([(X / Z) * f + cx, -(Y / Z) * f + cy])
