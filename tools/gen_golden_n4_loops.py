"""Fixtures for the SEARCH LOOPS of row N4: the Random Search and Coordinate Descent buttons of launch_smart_aligner (reference
utils/camera_estimation.py:606-650, :652-686), driven headlessly (tools/ref_widgets.py) on the stored Akbar grid with a seeded
np.random: the slider values the reference leaves behind and the IoU it found.  THIS CONTAINER ONLY (imports the reference)."""
import contextlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_widgets  # noqa: E402

ref_widgets.install()
import ref_import  # noqa: E402
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
import matplotlib  # noqa: E402
matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402
plt.show = lambda *a, **k: plt.close("all")

OUT = os.path.join(ROOT, "tests", "golden")
PC = cfg.PART_COLORS


def to_numpy(obj):
    if isinstance(obj, list):
        return np.array(obj, dtype=np.float32)
    if isinstance(obj, dict):
        return {k: to_numpy(v) for k, v in obj.items()}
    return obj


def sliders_now(S):
    return {k: float(S[k].value) for k in ("cam_x", "cam_y", "cam_z", "target_x", "target_y", "target_z", "f", "cx", "cy")}


cases = []
grid = np.load(os.path.join(OUT, "stored_Akbar_voxel_grid.npz"))["voxel_grid"]
cams = to_numpy(json.load(open(os.path.join(OUT, "stored_Akbar_camera_params_final.json"))))
front = mask_ingest.nearest_resize(mask_ingest.load_rgb("Akbar", "front"), int(np.max(grid.shape)))
base = cams["front"]
for (parts, lock, seed, rsteps, csteps, jitter) in (
        (["front_minarets", "back_minarets"], False, 5, 12, 3, (0.0, 0.0, 0.0)),
        (["front_minarets", "back_minarets"], False, 9, 25, 4, (35.0, -20.0, 60.0)),      # start away from the optimum: improvements happen
        (["full_building", "chhatris"], True, 3, 8, 2, (10.0, 10.0, -40.0)),
):
    init = {"cam_pos": base["cam_pos"].astype(np.float64) + np.array(jitter), "target": base["target"].astype(np.float64),
            "f": float(base["f"]) + jitter[0], "cx": float(base["cx"]) - jitter[1] / 2, "cy": float(base["cy"]) + 7.0}
    ref_widgets.install()                      # fresh registries (the module objects are re-used by the reference's imports)
    import ipywidgets  # noqa: F401
    ce.widgets = sys.modules["ipywidgets"]
    with contextlib.redirect_stdout(io.StringIO()):
        ce.launch_smart_aligner(grid, front, PC, parts_for_alignment=parts, init_params={k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in init.items()},
                                lock_xy_equal=lock)
    S = ref_widgets.CREATED["sliders"]; B = {b.description: b for b in ref_widgets.CREATED["buttons"]}
    start = sliders_now(S)
    S["Random Steps"].set_silently(rsteps); S["Coord Steps"].set_silently(csteps)
    np.random.seed(seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        B["Random Search"].click()
    after_random = sliders_now(S)
    log_r = [ln for ln in buf.getvalue().splitlines() if "Done" in ln]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        B["Coordinate Descent"].click()
    after_coord = sliders_now(S)
    log_c = [ln for ln in buf.getvalue().splitlines() if "Done" in ln]
    S["Powell MaxIter"].set_silently(2)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        B["Powell"].click()
    after_powell = sliders_now(S)
    log_p = [ln for ln in buf.getvalue().splitlines() if "Done" in ln]
    cases.append({"monument": "Akbar", "parts": parts, "lock_xy_equal": lock, "seed": seed, "random_steps": rsteps, "coord_steps": csteps,
                  "start": start, "after_random": after_random, "after_coord": after_coord, "log_random": log_r, "log_coord": log_c,
                  "powell_maxiter": 2, "after_powell": after_powell, "log_powell": log_p})
    print(json.dumps(cases[-1]["log_random"] + cases[-1]["log_coord"]), start != after_random, after_random != after_coord, flush=True)
np.savez_compressed(os.path.join(OUT, "n4_search_loops.npz"), front_Akbar=front)
json.dump({"cases": cases, "image": {"monument": "Akbar", "view": "front", "max_dim": int(np.max(grid.shape))}}, open(os.path.join(OUT, "n4_search_loops.json"), "w"), indent=1)
print("written", len(cases))
