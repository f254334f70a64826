"""The hand-off files either side of the path (row N3; host side, formats unchanged):

* voxel grids: `np.savez_compressed(path, voxel_grid=grid)` (reference notebook 1 cell 9, notebook 3 cell 9) read back with
  `np.load(path)["voxel_grid"]` (notebook 2 cell 3, notebook 3 cell 3) -- the stored layout is what partwise_carve returns,
  (D,H,W,3) uint8;
* camera parameters: `json.dump(to_json_safe(params))` (notebook 2 cell 11: arrays -> lists) and the `to_numpy` rule of notebook 3
  cell 3 on the way back: every JSON list becomes a float32 array, dicts recurse, scalars stay Python floats.
"""
import json

import numpy as np

__all__ = ["save_voxel_grid", "load_voxel_grid", "save_camera_params", "load_camera_params"]


def save_voxel_grid(path, grid):
    """grid: a NumPy array or a pb3d.device.DeviceGrid (downloaded once); the file is upstream's: one compressed array `voxel_grid`."""
    g = grid.numpy() if hasattr(grid, "numpy") and hasattr(grid, "buf") else np.asarray(grid)
    np.savez_compressed(path, voxel_grid=g)


def load_voxel_grid(path, on_device=False):
    """the `voxel_grid` array of a stored file; on_device=True: uploaded, as a pb3d.device.DeviceGrid"""
    with np.load(path) as f:
        g = f["voxel_grid"]
    if not on_device:
        return g
    from . import device as dev
    g = np.ascontiguousarray(g.astype(np.uint8, copy=False))
    return dev.DeviceGrid(dev.from_numpy(g), g.shape)


def _to_json_safe(obj):
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, dict):
        return {k: _to_json_safe(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_to_json_safe(v) for v in obj]
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def save_camera_params(path, params):
    with open(path, "w") as f:
        json.dump(_to_json_safe(params), f, indent=2)


def _to_numpy(obj):
    if isinstance(obj, list):
        return np.array(obj, dtype=np.float32)
    if isinstance(obj, dict):
        return {k: _to_numpy(v) for k, v in obj.items()}
    return obj


def load_camera_params(path):
    """a stored camera JSON as notebook 3 uses it: lists -> float32 arrays, scalars stay Python floats (which keeps
    project_colored_voxels on its float32 route under NumPy-2 promotion)"""
    with open(path) as f:
        return _to_numpy(json.load(f))
