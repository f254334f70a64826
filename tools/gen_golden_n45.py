"""Fixtures for rows N4 (camera objective of launch_smart_aligner, reference utils/camera_estimation.py:597-603, with
seg_img from mask_parts_from_image utils/mask_utils.py:89-97) and N5 (z-buffer visibility, reference
utils/eval_helpers_intra.py:134-190), captured from the live reference (THIS CONTAINER ONLY)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mask_ingest  # noqa: E402
import ref_import  # noqa: E402

vc, vu, pu, cg, ce, cfg = ref_import.load_reference()
import types  # noqa: E402
for name in ("tabulate",):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            m = types.ModuleType(name); m.tabulate = lambda *a, **k: ""; sys.modules[name] = m
import utils.mask_utils as mu  # noqa: E402
try:
    import utils.eval_helpers_intra as eh
except Exception as e:  # pragma: no cover
    print("eval_helpers_intra import failed:", repr(e)); raise
OUT = os.path.join(ROOT, "tests", "golden")
PC = cfg.PART_COLORS


def to_numpy(obj):
    if isinstance(obj, list):
        return np.array(obj, dtype=np.float32)
    if isinstance(obj, dict):
        return {k: to_numpy(v) for k, v in obj.items()}
    return obj


d = {}
meta = {}
for mon in ("Akbar", "Charminar"):
    grid = np.load(os.path.join(OUT, f"stored_{mon}_voxel_grid.npz"))["voxel_grid"]
    cams = to_numpy(json.load(open(os.path.join(OUT, f"stored_{mon}_camera_params_final.json"))))
    front = mask_ingest.nearest_resize(mask_ingest.load_rgb(mon, "front"), int(np.max(grid.shape)))
    H, W = front.shape[:2]
    parts = ["front_minarets", "back_minarets"]
    seg = mu.mask_parts_from_image(front, PC, parts)
    d[f"seg_{mon}"] = seg
    pts, cols = vu.get_voxel_points_by_parts(grid, PC, parts)
    sel = {p: PC[p] for p in parts}
    base = cams["front"]
    rng = np.random.default_rng(11)
    trials = []
    for t in range(6):
        p = {"cam_pos": base["cam_pos"].astype(np.float64) + rng.uniform(-1, 1, 3) * np.array([50, 50, 100]) * (t > 0),
             "target": base["target"].astype(np.float64) + rng.uniform(-1, 1, 3) * np.array([50, 50, 100]) * (t > 0),
             "f": float(base["f"]) + float(rng.uniform(-1, 1) * 50) * (t > 0), "cx": float(base["cx"]) + float(rng.uniform(-1, 1) * 20) * (t > 0),
             "cy": float(base["cy"]) + float(rng.uniform(-1, 1) * 20) * (t > 0)}
        proj = pu.project_colored_voxels(pts, cols, p["cam_pos"].copy(), p["target"].copy(), p["f"], p["cx"], p["cy"], H, W)
        _, iou = ce.compute_partwise_iou(proj, seg, sel)
        trials.append({"cam_pos": p["cam_pos"].tolist(), "target": p["target"].tolist(), "f": p["f"], "cx": p["cx"], "cy": p["cy"],
                       "neg_iou": float(-iou)})
    meta[f"objective_{mon}"] = {"parts": parts, "H": H, "W": W, "trials": trials, "npts": int(len(pts))}
    # N5: z-buffer of the whole object, visibility mask of one part (float32 camera from JSON, float64 variant)
    for mode in ("f32", "f64"):
        cam = {k: (v.astype(np.float64) if (mode == "f64" and isinstance(v, np.ndarray)) else v) for k, v in base.items()}
        cam = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
        if mon == "Charminar" and mode == "f64":
            continue
        zbuf = eh.compute_global_depth_buffer(grid, cam, H, W)
        ppts, _ = vu.get_voxel_points_by_parts(grid, PC, ["front_minarets"])
        vis = eh.project_part_visible(ppts, cam, zbuf, H, W)
        d[f"zbuf_{mon}_{mode}"] = zbuf
        d[f"vis_{mon}_{mode}"] = vis
        meta[f"zbuf_{mon}_{mode}"] = {"finite": int(np.isfinite(zbuf).sum()), "visible": int(vis.sum())}
np.savez_compressed(os.path.join(OUT, "n45_objective_zbuffer.npz"), **d)
json.dump(meta, open(os.path.join(OUT, "n45_objective_zbuffer.json"), "w"), indent=1)
print(json.dumps({k: (v if "zbuf" in k else [t["neg_iou"] for t in v["trials"]]) for k, v in meta.items()}, indent=1))
print("KB", os.path.getsize(os.path.join(OUT, "n45_objective_zbuffer.npz")) // 1024)
