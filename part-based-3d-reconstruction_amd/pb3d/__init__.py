"""pb3d -- MI355X-native semantic voxel carving & re-projection (host side).

Mirrors the reference's L2 function surface (utils.voxel_carving_utils, utils.voxel_utils,
utils.projection_utils, utils.camera_geometry, utils.camera_estimation.compute_partwise_iou,
utils.config) on top of libpb3d.so.  `install()` rebinds those names inside an imported
reference `utils` package so notebooks 1-3 run unchanged.
"""
from . import _hostmem, _lib, device, dist, formats, labels  # noqa: F401
from .formats import load_camera_params, load_voxel_grid, save_camera_params, save_voxel_grid  # noqa: F401
from .labels import (Palette, extrude_from_surface_labels, get_voxel_points_by_parts_labels, global_carve_labels, label_to_rgb,  # noqa: F401
                     left_right_guided_carve_labels, part_carve_labels, partwise_carve_labels, recolor_backward_components_labels, rgb_to_label,
                     voxel_grid_to_points_labels)
from ._hostmem import set_result_pool  # noqa: F401
from .camera_estimation import (CameraObjective, compute_partwise_iou, coordinate_descent, powell_search, projection_iou_by_part,  # noqa: F401
                                random_search)
from .eval_helpers_intra import compute_global_depth_buffer, project_part_visible  # noqa: F401
from .mask_utils import load_and_prepare_masks, load_mask, mask_parts_from_image  # noqa: F401
from .camera_geometry import look_at_rotation, project  # noqa: F401
from .deformation_estimation import (build_deformed_grid, deform_coords, deform_part, evaluate_part_deform,  # noqa: F401
                                     evaluate_part_deform_batch)
from .config import INTERIOR_PARTS, MAX_DIM, PART_COLORS, PART_COLORS_NP  # noqa: F401
from .projection_utils import project_colored_voxels  # noqa: F401
from .voxel_carving_utils import (apply_colored_mask_to_voxel_grid, carve_voxel_grid_with_masks, extrude_from_surface,  # noqa: F401
                                  global_carve, left_right_guided_carve, part_carve, partwise_carve, process_voxel_grid,
                                  recolor_backward_components)
from .voxel_utils import get_voxel_points_by_parts, voxel_grid_to_points  # noqa: F401

_PATCH = {
    "voxel_carving_utils": ["carve_voxel_grid_with_masks", "process_voxel_grid", "apply_colored_mask_to_voxel_grid",
                            "part_carve", "global_carve", "_occupancy", "left_right_guided_carve", "extrude_from_surface",
                            "recolor_backward_components", "partwise_carve"],
    "voxel_utils": ["get_voxel_points_by_parts", "voxel_grid_to_points"],
    "projection_utils": ["project_colored_voxels"],
    "camera_estimation": ["compute_partwise_iou"],
    "eval_helpers_intra": ["compute_global_depth_buffer", "project_part_visible"],
}


def install(utils_pkg=None):
    """Rebind the hot-path functions of an imported reference `utils` package to pb3d.

    Every module of the package that holds one of the names (the reference star-imports
    them across modules) is patched, so internal callers such as left_right_guided_carve or
    the camera aligner pick up the GPU path too.  Returns the list of (module, name) patched.
    """
    import importlib
    import sys
    import types

    if utils_pkg is None:
        utils_pkg = sys.modules.get("utils") or importlib.import_module("utils")
    here = sys.modules[__name__]
    patched = []
    names = {n: getattr(importlib.import_module(f"{__name__}.{mod}"), n) for mod, ns in _PATCH.items() for n in ns}
    for modname, mod in list(sys.modules.items()):
        if not isinstance(mod, types.ModuleType) or not (modname == utils_pkg.__name__ or modname.startswith(utils_pkg.__name__ + ".")):
            continue
        for n, fn in names.items():
            if n in mod.__dict__ and mod.__dict__[n] is not fn:
                mod.__dict__[n] = fn
                patched.append((modname, n))
    vis = sys.modules.get(utils_pkg.__name__ + ".visualization")
    if vis is not None and hasattr(vis, "plot_voxel"):
        here.voxel_carving_utils.plot_voxel = vis.plot_voxel
    return patched
