"""M5 (BASELINE configs[4]): the part-wise deformation re-projection loop on ALL FIVE monuments of the reference
(results/1 stored grids + results/2 final cameras, copied as data fixtures): golden digests captured by driving the
reference's notebook-3 widget closures headlessly (THIS CONTAINER ONLY; same machinery as tools/gen_golden_f8.py).
The "image" of every monument is the reference's own projection of its undeformed grid with the stored front camera."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_widgets  # noqa: E402

ref_widgets.install()
import ref_import  # noqa: E402

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


DEFORMS = [dict(scale_y=1.1, shift_y=3.0, scale_xz=0.9, shift_xz=2.0), dict(scale_y=0.85, shift_y=-6.0, scale_xz=1.2, shift_xz=-4.0),
           dict(scale_y=1.0, shift_y=-1.0, scale_xz=1.03, shift_xz=1.0)]
meta = {}
for mon in (os.environ.get("M5_ONLY", "Akbar,Bibi,Charminar,Itimad,Taj").split(",")):
    grid = np.load(os.path.join(OUT, f"stored_{mon}_voxel_grid.npz"))["voxel_grid"]
    cams = to_numpy(json.load(open(os.path.join(OUT, f"stored_{mon}_camera_params_final.json"))))
    cam = cams["front"]
    names = list(PC.keys())
    pts, cols = vu.get_voxel_points_by_parts(grid, PC, names)
    H, W = int(grid.shape[1]), int(grid.shape[2])
    image = pu.project_colored_voxels(pts, cols, cam["cam_pos"], cam["target"], cam["f"], cam["cx"], cam["cy"], H, W)
    ref_widgets.install()
    saved, store = de.launch_deform_viewer_fixed_camera(grid, PC, image=image, cam_params=cam, part_names=names)
    S = ref_widgets.CREATED["sliders"]; B = ref_widgets.CREATED["buttons"]
    save_btn, save_grid_btn = B[0], B[1]
    deform_coords = ref_widgets.closure_of(save_btn._clicks[0], "deform_coords")
    present = [n for n in names if len(vu.get_voxel_points_by_parts(grid, PC, [n])[0])]
    m = {"grid_shape": list(grid.shape), "image_shape": [H, W], "n_points": int(len(pts)), "points_sha256": sha(pts), "colors_sha256": sha(cols),
         "image_sha256": sha(image), "image_nonzero": int(image.any(-1).sum()), "cases": {}}
    for k, part in enumerate(present[:3]):
        dv = DEFORMS[k]
        S["Part"].set_silently(part)                  # no redraw per slider (each costs a projection of millions of points upstream);
        for kk, v in dv.items():                      # the save closure reads the slider values itself
            S[kk].set_silently(v)
        save_btn.click()
        coords, _ = vu.get_voxel_points_by_parts(grid, PC, [part])
        cd = deform_coords(coords.copy(), image.shape[:2], grid.shape[:3], dv)
        m["cases"][part] = {"deform": dv, "iou": saved[part]["iou"], "n_points": int(len(coords)), "n_deformed": int(len(cd)),
                            "coords_sha256": sha(cd.astype(np.int64))}
    save_grid_btn.click()
    full = store["grid"]
    m["deformed_grid_sha256"] = sha(full); m["deformed_grid_occupied"] = int(np.any(full > 0, -1).sum())
    meta[mon] = m
    print(mon, m["grid_shape"], m["n_points"], {p: c["iou"] for p, c in m["cases"].items()}, flush=True)
# one monument per process (the reference needs minutes per monument): partial files are merged when all five exist
part_dir = os.path.join(ROOT, "gpurun_out", "m5_parts"); os.makedirs(part_dir, exist_ok=True)
for mon, m in meta.items():
    json.dump(m, open(os.path.join(part_dir, f"{mon}.json"), "w"), indent=1)
have = {f[:-5]: json.load(open(os.path.join(part_dir, f))) for f in sorted(os.listdir(part_dir)) if f.endswith(".json")}
if all(k in have for k in ("Akbar", "Bibi", "Charminar", "Itimad", "Taj")):
    json.dump({k: have[k] for k in ("Akbar", "Bibi", "Charminar", "Itimad", "Taj")}, open(os.path.join(OUT, "m5_five_monuments_deformation.json"), "w"), indent=1)
    print("merged", flush=True)
