import contextlib, io, json, os, sys, time
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, os.path.join(ROOT, "part-based-3d-reconstruction_amd"))
import pb3d
from pb3d import device as dev
from pb3d import voxel_carving_utils as V
g = {k: v for k, v in np.load(os.path.join(ROOT, "tests", "golden", "f9_Taj_512_masks.npz")).items()}
jobs = [(["full_building"], 90), (["chhatris"], 90), (["plinth"], 90), (["front_minarets"], 90), (["small_minarets"], 90), (["dome"], 90)]
d_gc = pb3d.global_carve(g["binary"], g["ext"], angle_interval=90, on_device=True)
d_pc = pb3d.part_carve(d_gc, g["ext"], jobs)
W, H, D, _ = d_pc.shape
PCN = pb3d.PART_COLORS_NP
sm = np.asarray(g["ext"]); key = V._color_key(sm)
for part, ang in (("dome", 5), ("front_minarets", 5), ("chhatris", 45)):
    mask2d = V._is_color(sm, key, PCN[part])
    row = {"part": part}
    for abl in (0, 1, 2, 4, 7):
        pb3d._lib.set_tuning("crop_ablate", abl)
        ts = []
        for r in range(4):
            d_w = dev.DeviceBuffer(W * H * D * 3)
            pb3d._lib.check(pb3d._lib.load().pb3d_d2d(pb3d._lib.ctx(), d_w._as_void() if hasattr(d_w, "_as_void") else __import__("ctypes").c_void_p(d_w.ptr), __import__("ctypes").c_void_p(d_pc.buf.ptr), W * H * D * 3))
            dev.sync(); t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                V._lrgc_dev(d_w, (W, H, D), mask2d, PCN[part], ang)
            dev.sync(); ts.append(time.perf_counter() - t0)
            d_w.free()
        row[f"abl{abl}_ms"] = round(min(ts) * 1e3, 3)
    pb3d._lib.set_tuning("crop_ablate", 0)
    print(json.dumps(row), flush=True)
