"""F8: golden vectors of the notebook-3 deformation closures (reference utils/deformation_estimation.py:70-98,
:100-146, :262-313), captured by driving launch_deform_viewer_fixed_camera headlessly (THIS CONTAINER ONLY)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_widgets  # noqa: E402

ref_widgets.install()
import ref_import  # noqa: E402  (its ipywidgets/IPython stubs are skipped: already registered above)
import mask_ingest  # noqa: E402

sys.modules.setdefault("cv2", None)
_orig_stub = ref_import._stub


def _keep_widgets(name, **attrs):
    if name in ("ipywidgets", "IPython", "IPython.display"):
        return sys.modules[name]
    return _orig_stub(name, **attrs)


ref_import._stub = _keep_widgets
del sys.modules["cv2"]
vc, vu, pu, cg, ce, cfg = ref_import.load_reference()
import matplotlib.pyplot as plt  # noqa: E402
plt.show = lambda *a, **k: None
import utils.deformation_estimation as de  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PC = cfg.PART_COLORS
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def to_numpy(obj):
    if isinstance(obj, list):
        return np.array(obj, dtype=np.float32)
    if isinstance(obj, dict):
        return {k: to_numpy(v) for k, v in obj.items()}
    return obj


grid = np.load(os.path.join(OUT, "stored_Akbar_voxel_grid.npz"))["voxel_grid"]
cams = to_numpy(json.load(open(os.path.join(OUT, "stored_Akbar_camera_params_final.json"))))
front = mask_ingest.nearest_resize(mask_ingest.load_rgb("Akbar", "front"), int(np.max(grid.shape)))

saved, store = de.launch_deform_viewer_fixed_camera(grid, PC, image=front, cam_params=cams["front"], part_names=list(PC.keys()))
S = ref_widgets.CREATED["sliders"]; B = ref_widgets.CREATED["buttons"]
save_btn, save_grid_btn = B[0], B[1]
deform_coords = ref_widgets.closure_of(save_btn._clicks[0], "deform_coords")

cases = {"front_minarets": dict(scale_y=1.1, shift_y=3.0, scale_xz=0.9, shift_xz=2.0),
         "chhatris": dict(scale_y=0.8, shift_y=-5.0, scale_xz=1.25, shift_xz=-3.0),
         "full_building": dict(scale_y=1.0, shift_y=-1.0, scale_xz=1.02, shift_xz=0.0),
         "windows": dict(scale_y=2.0, shift_y=40.0, scale_xz=2.0, shift_xz=30.0)}      # pushes voxels out of the grid
d = {"front_mask": front}
meta = {"voxel_shape": list(grid.shape[:3]), "image_shape": list(front.shape[:2]), "cases": {}}
for part, dv in cases.items():
    S["Part"].value = part
    for k, v in dv.items():
        S[k].value = v
    save_btn.click()
    coords, colors = vu.get_voxel_points_by_parts(grid, PC, [part])
    cd = deform_coords(coords.copy(), front.shape[:2], grid.shape[:3], dv)
    meta["cases"][part] = {"deform": dv, "iou": saved[part]["iou"], "n_points": int(len(coords)), "n_deformed": int(len(cd)),
                           "coords_sha256": sha(cd.astype(np.int64)), "dtype": str(cd.dtype)}
    if part in ("chhatris", "windows"):
        d[f"coords_{part}"] = cd.astype(np.int32)
save_grid_btn.click()
full = store["grid"]
meta["deformed_grid_sha256"] = sha(full); meta["deformed_grid_occupied"] = int(np.any(full > 0, -1).sum())
d["deformed_grid"] = full
# a small random cloud through the raw closure (non-uniform colours never reach it; coordinates are voxel indices)
rng = np.random.default_rng(5)
pts = rng.integers(0, 40, (500, 3)).astype(np.float32)
dv = dict(scale_y=1.37, shift_y=7.0, scale_xz=0.61, shift_xz=-9.0)
d["rand_pts"] = pts
d["rand_coords"] = deform_coords(pts.copy(), (90, 120), (40, 44, 48), dv).astype(np.int32)
meta["rand"] = {"deform": dv, "image_shape": [90, 120], "voxel_shape": [40, 44, 48]}
np.savez_compressed(os.path.join(OUT, "f8_deformation.npz"), **d)
json.dump(meta, open(os.path.join(OUT, "f8_deformation.json"), "w"), indent=1)
print(json.dumps(meta, indent=1)[:1500])
print("size KB", os.path.getsize(os.path.join(OUT, "f8_deformation.npz")) / 1024)
