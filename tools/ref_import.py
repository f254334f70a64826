"""Import harness for the upstream reference (THIS CONTAINER ONLY).

Used solely by tools/gen_golden.py and tools/pin_oracle.py to run the reference's
own Python (/root/reference, read-only) so that golden input/output vectors can
be captured under tests/golden/.  Nothing in tests/, bench.py or the product
imports this module at run time on the GPU box (the reference does not travel).

The reference needs five third-party modules that are absent from this image
(cv2, skimage, trimesh, ipywidgets, IPython); none of them is touched by the
hot-path functions, so empty stand-in modules are registered before import.
"""
import sys
import types

sys.dont_write_bytecode = True  # /root/reference is read-only

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _unavailable(*a, **k):
    raise RuntimeError("stubbed third-party function called")


def load_reference():
    """Returns (voxel_carving_utils, voxel_utils, projection_utils, camera_geometry,
    camera_estimation, config) modules of the reference."""
    import matplotlib
    matplotlib.use("Agg")
    if "cv2" not in sys.modules:
        _stub("cv2", imread=_unavailable, cvtColor=_unavailable, resize=_unavailable,
              COLOR_BGR2RGB=4, INTER_NEAREST=0)
        sk = _stub("skimage")
        sk.measure = _stub("skimage.measure", marching_cubes=_unavailable,
                           regionprops=_unavailable, label=_unavailable)
        _stub("trimesh", Trimesh=_unavailable)
        _stub("ipywidgets")
        ip = _stub("IPython", get_ipython=lambda: None, version_info=(8, 12, 3))
        ip.display = _stub("IPython.display", display=_unavailable, clear_output=_unavailable)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import utils.voxel_carving_utils as vc
    import utils.voxel_utils as vu
    import utils.projection_utils as pu
    import utils.camera_geometry as cg
    import utils.camera_estimation as ce
    import utils.config as cfg
    return vc, vu, pu, cg, ce, cfg
